// Process-wide options of libbya_hip.so: set through bya_set_option (include/bya.h), read with bya_opt() where a launch is
// prepared.  The entry points never look at the environment -- the Python host reads its BYA_* variables ONCE when the
// library is loaded (ops.apply_env_options) and hands them over through the setter, so the C ABI stays a function of
// its arguments plus this one table.
#pragma once
#include <atomic>
#include <stdint.h>
#include "../../include/bya.h"

extern std::atomic<int32_t> g_bya_options[BYA_OPT_COUNT];      // defined in misc.hip

inline int32_t bya_opt(int key) { return g_bya_options[key].load(std::memory_order_relaxed); }
// one of the "reference form" bits of BYA_OPT_REFERENCE_FORMS (tests: the older kernel form of a closed A/B, bit-identical)
inline bool bya_ref_form(int32_t bit) { return (bya_opt(BYA_OPT_REFERENCE_FORMS) & bit) != 0; }
