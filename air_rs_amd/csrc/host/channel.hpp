// channel.hpp -- unbounded multi-producer single-consumer FIFO with the semantics thread 2 of the
// reference relies on (std::sync::mpsc, src/adsb.rs:131,146): send() fails once the receiver is
// gone (adsb.rs:108-111 prints and returns), recv() fails once every sender is gone and the queue
// is drained (adsb.rs:95 ends the `while let Ok(..)` loop).
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <utility>

namespace air_rs_amd {

template <typename T> struct ChannelState {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    int senders = 0;
    bool receiver_alive = true;
};

template <typename T> class Sender {
public:
    Sender() = default;
    explicit Sender(std::shared_ptr<ChannelState<T>> s) : st_(std::move(s))
    {
        std::lock_guard<std::mutex> g(st_->mu);
        st_->senders++;
    }
    Sender(const Sender &o) : st_(o.st_)
    {
        if (st_) {
            std::lock_guard<std::mutex> g(st_->mu);
            st_->senders++;
        }
    }
    Sender(Sender &&o) noexcept : st_(std::move(o.st_)) {}
    Sender &operator=(Sender o)
    {
        std::swap(st_, o.st_);
        return *this;
    }
    ~Sender() { drop(); }
    // `drop(tx)` (adsb.rs:88,121)
    void drop()
    {
        if (!st_) return;
        {
            std::lock_guard<std::mutex> g(st_->mu);
            st_->senders--;
        }
        st_->cv.notify_all();
        st_.reset();
    }
    // false == Err(SendError): the receiver was dropped
    bool send(T v)
    {
        if (!st_) return false;
        {
            std::lock_guard<std::mutex> g(st_->mu);
            if (!st_->receiver_alive) return false;
            st_->q.push_back(std::move(v));
        }
        st_->cv.notify_one();
        return true;
    }

private:
    std::shared_ptr<ChannelState<T>> st_;
};

template <typename T> class Receiver {
public:
    Receiver() = default;
    explicit Receiver(std::shared_ptr<ChannelState<T>> s) : st_(std::move(s)) {}
    Receiver(const Receiver &) = delete; // mpsc::Receiver is not Clone
    Receiver(Receiver &&o) noexcept : st_(std::move(o.st_)) {}
    Receiver &operator=(Receiver &&o) noexcept
    {
        drop();
        st_ = std::move(o.st_);
        return *this;
    }
    ~Receiver() { drop(); }
    void drop()
    {
        if (!st_) return;
        {
            std::lock_guard<std::mutex> g(st_->mu);
            st_->receiver_alive = false;
            st_->q.clear();
        }
        st_.reset();
    }
    // nullopt == Err(RecvError): all senders dropped and nothing queued
    std::optional<T> recv()
    {
        std::unique_lock<std::mutex> g(st_->mu);
        st_->cv.wait(g, [&] { return !st_->q.empty() || st_->senders == 0; });
        if (st_->q.empty()) return std::nullopt;
        T v = std::move(st_->q.front());
        st_->q.pop_front();
        return v;
    }

    // `try_recv()`: 0 = Ok(v) (written to *out), 1 = Err(Empty), 2 = Err(Disconnected)
    int try_recv(T *out)
    {
        std::lock_guard<std::mutex> g(st_->mu);
        if (!st_->q.empty()) {
            *out = std::move(st_->q.front());
            st_->q.pop_front();
            return 0;
        }
        return st_->senders == 0 ? 2 : 1;
    }

private:
    std::shared_ptr<ChannelState<T>> st_;
};

// `mpsc::channel()`
template <typename T> std::pair<Sender<T>, Receiver<T>> channel()
{
    auto st = std::make_shared<ChannelState<T>>();
    return {Sender<T>(st), Receiver<T>(st)};
}

} // namespace air_rs_amd
