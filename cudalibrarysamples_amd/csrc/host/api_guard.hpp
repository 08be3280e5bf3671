// api_guard.hpp — the exception barrier of the C ABI.
// The host side is C++17 with std::vector / std::map / std::string / std::shared_ptr behind every entry point; the callers are C
// programs, ctypes and the reference's samples (utils.cuh:35-46 turns a STATUS into an exception on ITS side).  No exception may
// cross the boundary: every `extern "C"` function of libcutensor.so / libcutensorMg.so /
// libcutensorMp.so is a function-try-block that ends in one of these handlers.  An allocation failure becomes
// CUTENSOR_STATUS_ALLOC_FAILED, anything else CUTENSOR_STATUS_INTERNAL_ERROR; diagnostics that return int / size_t / pointers
// answer -1 / 0 / nullptr.  tests/test_abi.py::test_allocation_failure_* drives the entry points with a throwing operator new.
#pragma once
#include <new>

#include <cutensor.h>

#define CTAMD_API_CATCH                                                          \
    catch (const std::bad_alloc&) { return CUTENSOR_STATUS_ALLOC_FAILED; }       \
    catch (...) { return CUTENSOR_STATUS_INTERNAL_ERROR; }
#define CTAMD_API_CATCH_INT  catch (...) { return -1; }
#define CTAMD_API_CATCH_ZERO catch (...) { return 0; }
#define CTAMD_API_CATCH_NULL catch (...) { return nullptr; }
#define CTAMD_API_CATCH_VOID catch (...) { }

// Behaviour-changing environment switches that exist for the test-suite and the measurement tools (CUTENSOR_AMD_H16_WAVES, _GEN, _NT,
// _PEEL, _FUSED_FOLD, _H16P, _H16P_GRID, _H16_STRIPS, CUTENSORMG_AMD_{ASSUME_RCCL,DIRECT,PEEL,QSPLIT,SHARD2,TEST_DROP_WAITS},
// CUTENSORMP_AMD_ALGO) are read only by the TEST-HOOKS flavour of the libraries (make HOOKS=1 -> lib_hooks/, what tests/ loads) and by
// research builds.  In the production libraries (lib/) the macro is a null pointer and the names do not even appear as strings:
// `strings lib/libcutensor*.so | grep CUTENSOR` lists CUTENSOR_LOG_LEVEL and the documented user switches of cuTENSORMg
// (CUTENSORMG_AMD_FORCE_GATHER / _TRANSPORT / _WAVES) only.
#include <cstdlib>
#if defined(CTAMD_TEST_HOOKS) || defined(CTAMD_RESEARCH_KERNELS)
#define CTAMD_HOOK_ENV(NAME) std::getenv(NAME)
#define CTAMD_HOOKS_BUILT 1
#else
#define CTAMD_HOOK_ENV(NAME) (static_cast<const char*>(nullptr))
#define CTAMD_HOOKS_BUILT 0
#endif
