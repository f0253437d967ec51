// jn_elas_shim.hpp — drop-in `class Elas` over libjn_stereo.so.
//
// Including this header instead of the reference's "elas.h" makes the call site in
// src/obstacle_avoidance/point_cloud.cpp:416-419 compile unchanged:
//
//     Elas::parameters param;                 // elas.h:85 (ROBOTICS defaults)
//     param.postprocess_only_left = true;
//     Elas elas(param);
//     elas.process(left.data, right.data, leftdpf.ptr<float>(0), rightdpf.ptr<float>(0), dims);
//
// The reference constructs a fresh Elas per frame; creating GPU buffers per frame would be wasteful,
// so the shim keeps one jn_elas handle per (parameters, width, height) alive for the process.
#pragma once
#include <stdint.h>
#include <string.h>
#include <iostream>
#include "jn_stereo.h"

class Elas {
 public:
  enum setting { ROBOTICS = JN_SETTING_ROBOTICS, MIDDLEBURY = JN_SETTING_MIDDLEBURY };

  struct parameters : public jn_elas_params {            // same field names as elas.h:60-82
    parameters(setting s = ROBOTICS) { jn_elas_params_default(this, (int32_t)s); }
  };

  explicit Elas(parameters param) : param_(param) {}
  ~Elas() {}

  // dims = {width, height, bytes per line}; D1/D2 caller-allocated width*height floats, (width/2)*(height/2) with param.subsampling (elas.h:154-162)
  void process(uint8_t* I1, uint8_t* I2, float* D1, float* D2, const int32_t* dims) {
    jn_elas* h = handle(dims[0], dims[1]);
    if (!h) { std::cout << "ERROR: jn_elas_create failed (no MI355X / unsupported parameters)" << std::endl; return; }
    jn_elas_process(h, I1, I2, D1, D2, dims);              // prints the reference's message itself on <3 support points
  }

 private:
  parameters param_;
  jn_elas* handle(int32_t w, int32_t h) {
    struct Cache { jn_elas_params p; int32_t w, h; jn_elas* handle; };
    static Cache c = {{}, 0, 0, nullptr};
    if (c.handle && c.w == w && c.h == h && memcmp(&c.p, static_cast<jn_elas_params*>(&param_), sizeof(jn_elas_params)) == 0)
      return c.handle;
    if (c.handle) { jn_elas_destroy(c.handle); c.handle = nullptr; }
    jn_elas* out = nullptr;
    if (jn_elas_create(&param_, w, h, /*max_batch*/ 1, /*device*/ 0, /*host_threads*/ 8, /*slots*/ 1, &out) != JN_OK) return nullptr;
    c.p = param_; c.w = w; c.h = h; c.handle = out;
    return out;
  }
};
