/*
 * aej_testing.h -- TEST-ONLY entry points of libaejpeg_hip.so.  Not part of the drop-in boundary (include/aej.h): nothing a caller of
 * the reference's API would bind.  Used by tests/ to reach error paths that cannot be provoked from outside.
 */
#ifndef AEJ_TESTING_H
#define AEJ_TESTING_H

#include "aej.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The next whole-path call (aej_encode_batch / _begin) fails with AEJ_ERR_STATE right after it has enqueued stage `stage` (an AEJ_STAGE_*
 * value; -1 disarms) of its first part, i.e. with work in flight -- the error paths must drain it.  One-shot. */
AEJ_API int aej_test_fail_after_stage(aej_ctx *ctx, int stage);

#ifdef __cplusplus
}
#endif
#endif /* AEJ_TESTING_H */
