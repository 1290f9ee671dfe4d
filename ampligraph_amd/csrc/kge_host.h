// Host-side helpers shared by the translation units of libamdkge: error reporting that never
// throws across the C ABI, argument validation, per-model constants.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <exception>
#include <new>

#include "../../include/amdkge.h"
#include "kge_device.h"

namespace kge {

int set_error(int code, const char* msg);            // stores a thread-local message, returns code
int set_error_hip(hipError_t e, const char* where);  // AMDKGE_EHIP with hipGetErrorString
int check_launch(const char* kernel_name);           // hipGetLastError() after a launch

// "Once" for hipFuncSetAttribute-style set-up: a function's attributes belong to (function, DEVICE), and a process may drive several
// devices (session groups: one replica and one host thread per GPU), so a process-wide `static bool` would set up device 0 only.
// One bit per device ordinal (mod 64); racing threads at worst both do the idempotent set-up.
struct PerDeviceOnce {
    unsigned long long mask = 0ull;
    int dev = 0;
    bool need() {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        return !((__atomic_load_n(&mask, __ATOMIC_RELAXED) >> (dev & 63)) & 1ull);
    }
    void done() { __atomic_fetch_or(&mask, 1ull << (dev & 63), __ATOMIC_RELAXED); }
};

// The session layers allocate on the host (staging vectors, the per-device threads of a group, the registries): those can throw,
// the C ABI cannot.  Their entry points are function-try-blocks closed by KGE_CATCH("name").
#define KGE_CATCH(NAME)                                                                                              \
    catch (const std::bad_alloc&) { return kge::set_error(AMDKGE_ENOMEM, NAME ": out of host memory"); }             \
    catch (const std::exception& e) { return kge::set_error(AMDKGE_EINVAL, e.what()); }                              \
    catch (...) { return kge::set_error(AMDKGE_EINVAL, NAME ": unexpected C++ exception"); }

inline int validate_model(const amdkge_model* m) {
    if (!m) return set_error(AMDKGE_EINVAL, "model descriptor is NULL");
    if (m->scoring_type < AMDKGE_TRANSE || m->scoring_type > AMDKGE_ROTATE)
        return set_error(AMDKGE_EINVAL, "unknown scoring_type (expected TransE/DistMult/ComplEx/HolE/RotatE)");
    if (m->k <= 0) return set_error(AMDKGE_EINVAL, "k must be positive");
    if (m->k_pad != 0 && m->k_pad < m->k) return set_error(AMDKGE_EINVAL, "k_pad must be 0 (dense rows) or >= k");
    if (m->k_full != 0 && m->k_full < m->k) return set_error(AMDKGE_EINVAL, "k_full must be 0 (the model is whole) or >= k (the model is a column slice of a k_full-unit model)");
    if (m->n_ents <= 0 || m->n_rels <= 0)
        return set_error(AMDKGE_EINVAL, "entity / relation table sizes must be positive (model not built?)");
    if (m->n_ents > 0x7FFFFFFFll || m->n_rels > 0x7FFFFFFFll)
        return set_error(AMDKGE_EINVAL, "row ids are int32: tables are limited to 2^31-1 rows");
    return AMDKGE_OK;
}

// units per half AS STORED (include/amdkge.h "STORED row layout") and floats per stored row
inline int stored_k(const amdkge_model* m) { return m->k_pad > 0 ? m->k_pad : m->k; }
inline int row_floats(const amdkge_model* m) { return internal_k_of(m->scoring_type, stored_k(m)); }

inline ModelConst model_const(const amdkge_model* m) {
    ModelConst mc;
    mc.score_scale = 1.f;
    mc.score_sign = 1.f;
    mc.phase_div = 1.f;
    if (m->scoring_type == AMDKGE_TRANSE || m->scoring_type == AMDKGE_ROTATE) mc.score_sign = -1.f;
    // (a column slice -- amdkge_model.k_full -- scores with the constants of the WHOLE model: its sums are partial sums of that one)
    const double kf = m->k_full > 0 ? (double)m->k_full : (double)m->k;
    if (m->scoring_type == AMDKGE_HOLE) mc.score_scale = (float)(2.0 / kf);  // HolE.py:45
    if (m->scoring_type == AMDKGE_ROTATE) {
        const double R = m->max_rel_size > 0 ? (double)m->max_rel_size : 1.0;           // RotatE.py:87-94
        const double embedding_range = sqrt(6.0 / (2.0 * kf * R));                      // RotatE.py:95
        mc.phase_div = (float)(embedding_range / 3.14159265358979323846);               // :96 (pi from math)
    }
    return mc;
}

}  // namespace kge
