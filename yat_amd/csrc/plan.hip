// Launch-plan replay (include/yat_hip.h, "launch plans"): one C call walks a recorded list of entry-point calls and
// stream / event operations.  The per-entry-point trampolines are generated from the header at build time
// (yat_amd/build.py -> build/plan_dispatch.inc): each unpacks the uniform argument record into the typed call.
#include "common.hpp"
#include <string.h>

#include "../build/plan_dispatch.inc"      // kPlanNames[], kPlanCount, plan_dispatch(op, a)

extern "C" {

int yat_plan_op_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < kPlanCount; ++i)
        if (!strcmp(kPlanNames[i], name)) return i;
    return -1;
}

int yat_plan_replay(const yat_plan_entry* e, int n, int* failed_index) {
    if (!e || n < 0) return YAT_EINVAL;
    for (int i = 0; i < n; ++i) {
        const yat_plan_entry& x = e[i];
        int rc;
        if (x.op >= 0) rc = x.op < kPlanCount ? plan_dispatch(x.op, x.a) : YAT_EINVAL;
        else if (x.op == YAT_PLAN_EVENT_RECORD) rc = (int)hipEventRecord((hipEvent_t)x.a[0].p, (hipStream_t)x.a[1].p);
        else if (x.op == YAT_PLAN_STREAM_WAIT_EVENT) rc = (int)hipStreamWaitEvent((hipStream_t)x.a[0].p, (hipEvent_t)x.a[1].p, 0);
        else rc = YAT_EINVAL;
        if (rc) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return YAT_OK;
}

}  // extern "C"
