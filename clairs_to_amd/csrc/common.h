// Shared host-side helpers for the C-ABI implementation (error reporting, HIP call checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <exception>
#include <new>
#include "../../include/clairsto_amd.h"

namespace cto {

void set_error(const char* fmt, ...);

#define CTO_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::cto::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return CTO_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define CTO_REQUIRE(cond, code, ...)                \
    do {                                            \
        if (!(cond)) {                              \
            ::cto::set_error(__VA_ARGS__);          \
            return (code);                          \
        }                                           \
    } while (0)

// Closes a function-try-block of a C-ABI entry point: no C++ exception crosses the boundary (`extern "C" int f(...) try { ... } CTO_CATCH("f", int)`)
#define CTO_CATCH(name, T)                                                  \
    catch (const std::bad_alloc&) {                                         \
        ::cto::set_error("%s: out of memory", name);                        \
        return T(CTO_ENOMEM);                                               \
    } catch (const std::exception& e) {                                     \
        ::cto::set_error("%s: %s", name, e.what());                         \
        return T(CTO_EINVAL);                                               \
    }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace cto
