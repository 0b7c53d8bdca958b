// TEST SCAFFOLDING (oracle/_ref build): mutex queue standing in for moodycamel::ConcurrentQueue,
// needed only because the dead header include/readsqueue.hpp is still #included by the reference.
#pragma once
#include <queue>
#include <mutex>
namespace moodycamel {
template <typename T> class ConcurrentQueue {
  std::queue<T> q; std::mutex m;
public:
  explicit ConcurrentQueue(size_t = 0) {}
  bool try_enqueue(const T& v) { std::lock_guard<std::mutex> l(m); q.push(v); return true; }
  bool try_dequeue(T& v) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; v = q.front(); q.pop(); return true; }
  size_t size_approx() { std::lock_guard<std::mutex> l(m); return q.size(); }
};
}
