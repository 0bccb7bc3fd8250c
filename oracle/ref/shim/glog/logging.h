// Minimal stand-in for glog so that /root/reference/src/helpers.cpp compiles
// verbatim where it lies (glog itself needs CMake-generated headers; the
// reference's build system is not run - see oracle/ref/Makefile).
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
namespace xivo_ref_shim {
struct NullStream {
  template <class T> NullStream& operator<<(const T&) { return *this; }
  NullStream& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
struct FatalStream {
  std::ostringstream os;
  template <class T> FatalStream& operator<<(const T& v) { os << v; return *this; }
  ~FatalStream() { std::cerr << "CHECK failed: " << os.str() << std::endl; std::abort(); }
};
}  // namespace xivo_ref_shim
#define LOG(sev) ::xivo_ref_shim::NullStream()
#define VLOG(n) ::xivo_ref_shim::NullStream()
#define CHECK(cond) if (cond) {} else ::xivo_ref_shim::FatalStream() << #cond << " "
