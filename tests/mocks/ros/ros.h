// declaration-only stand-in for roscpp (see tests/mocks/README.md)
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <boost_shared_ptr_stub.h>
#include <std_msgs/Header.h>
namespace ros {
struct WallDuration { double toSec() const; };
struct WallTime { static WallTime now(); WallDuration operator-(const WallTime&) const; };
class Publisher { public: template <class M> void publish(const M&) const {} };
class Subscriber {};
class NodeHandle {
 public:
  NodeHandle();
  template <class M> Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
  template <class M, class T> Subscriber subscribe(const std::string&, uint32_t, void (T::*)(const boost::shared_ptr<M const>&), T*) { return Subscriber(); }
  bool getParam(const std::string& key, double& d) const;
  bool getParam(const std::string& key, int& i) const;
  bool getParam(const std::string& key, std::string& s) const;
};
void init(int& argc, char** argv, const std::string& name, uint32_t options = 0);
void spin();
void shutdown();
bool ok();
}  // namespace ros
#define ROS_FATAL(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_INFO(...) std::fprintf(stderr, __VA_ARGS__)
