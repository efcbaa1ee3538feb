// wdf_capi_common.h -- shared by the translation units of libwdf_hip.so: error string, launch
// check, the one-shot event bracket.  No algorithm lives here.
#pragma once
#include <atomic>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/wdf_hip.h"

namespace wdfcapi {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(WDF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return WDF_OK;
}

// wdf_event_bracket_next(): events to record immediately before / after the next RECURRENCE
// kernel the library launches (the forward or reverse sweep itself, not the verify / combine /
// reduce helpers that share its C call), so a harness can time exactly the kernel rocprofv3
// reports.  One-shot, process-wide: the call that consumes it may come from another thread
// than the one that armed it (torch's autograd engine runs a backward on its own thread).
struct EventPair { hipEvent_t e0, e1; };
extern std::atomic<EventPair*> g_bracket;          // armed pair or null: ONE word, so a start never goes without its stop

struct EventBracket {
    hipStream_t s;
    EventPair* p;
    explicit EventBracket(hipStream_t stream) : s(stream), p(g_bracket.exchange(nullptr))
    {
        if (p && p->e0) (void)hipEventRecord(p->e0, s);
    }
    ~EventBracket()
    {
        if (p) {
            if (p->e1) (void)hipEventRecord(p->e1, s);
            delete p;
        }
    }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

}  // namespace wdfcapi
