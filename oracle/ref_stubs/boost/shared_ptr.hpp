// ORACLE (test infrastructure only).  NOT boost: the names the reference's plugin source uses (shared_ptr / make_shared / bind with the global _1 _2 placeholders /
// mutex::scoped_lock / replace_all), mapped onto the standard library
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <string>
namespace boost {
using std::make_shared;
using std::shared_ptr;
template <class... A> auto bind(A&&... a) { return std::bind(std::forward<A>(a)...); }
class mutex {
 public:
    struct scoped_lock { explicit scoped_lock(mutex& m) : _l(m._m) {} std::lock_guard<std::mutex> _l; };
 private:
    std::mutex _m;
};
inline void replace_all(std::string& s, const std::string& from, const std::string& to) {
    for (size_t p = 0; !from.empty() && (p = s.find(from, p)) != std::string::npos; p += to.size()) s.replace(p, from.size(), to);
}
}  // namespace boost
using std::placeholders::_1;
using std::placeholders::_2;
