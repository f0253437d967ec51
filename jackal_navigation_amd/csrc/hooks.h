// hooks.h — the library's environment switches come in two kinds.
//   getenv("JN_X")       a switch that SHIPS: it selects between routes that give the same results (or sets a time-out, a library path); every
//                        one of them is listed with its default in INTEGRATION.md ("Environment switches").
//   JN_HOOK_ENV("JN_X")  a test hook, a profiling switch (several make results WRONG: JN_DENSE_DBG, JN_OWNER_DBG, JN_SGM_DBG, JN_SGM_EXP) or an
//                        A/B knob of a measurement script.  Compiled only into the hooks build (make hooks: -DJN_HOOKS ->
//                        libjn_stereo_hooks.so, what the tests and scripts that need them load through JN_STEREO_LIB / hooks_library()); in the
//                        release library the call is the constant nullptr, the name is not in the binary, and the kernels carry neither the
//                        argument nor the branches (JN_DBG_*).
#pragma once
#include <cstdlib>
#ifdef JN_HOOKS
#define JN_HOOK_ENV(name) getenv(name)
#else
#define JN_HOOK_ENV(name) (static_cast<const char*>(nullptr))
#endif
