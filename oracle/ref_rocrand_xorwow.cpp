// oracle/ref_rocrand_xorwow.cpp — TEST INFRASTRUCTURE, not part of the product.
//
// cuRAND (curand_kernel.h, CUDA Toolkit 12.2) is a dependency of the reference that is neither in its tree nor in this
// image, so the XORWOW generator behind `curand_init(seed, pixel_index, 0, &state[pixel_index])` (core/deviceManage.h:127)
// and `curand_uniform` (core/render.h:116-117, material/material.h:14-16) is restated from the published algorithm.
// The image DOES ship an independent third-party XORWOW: rocRAND's `rocrand_device::xorwow_engine`
// (/opt/rocm/include/rocrand/rocrand_xorwow.h, `__host__ __device__` throughout, jump matrices precomputed by its
// authors in rocrand_xorwow_precomputed.h).  It has
//   * the same recurrence (Marsaglia xorwow: five 32-bit words + the Weyl sequence d += 362437, output d + x[4]),
//   * the same stream structure (subsequence k = 2^67 * k steps ahead, d unchanged; offsets as powers of the same matrix),
//   * DIFFERENT seed-scrambling constants and a different uniform mapping (so curand_init's constants and
//     curand_uniform's `x * 2^-32 + 2^-33` stay PARITY UNPINNED).
// This file exposes the engine on RAW states (words and d handed in by the caller, no seeding involved), compiled for
// the HOST with the image's HIP front end exactly as it is; nothing is stood in for, nothing is copied:
//     clang++ -x hip --cuda-host-only -no-hip-rt -O2
// tests/test_ref_xorwow.py holds the oracle's step (o_curand, crt_oracle.c) and its 2^67-step and offset skips (xw_skip:
// GF(2) matrix powers built from the step itself by repeated squaring) to it, through the committed fixture
// tests/golden/rocrand_xorwow_vectors.npz (tests/make_xorwow_fixtures.py).  That pins the part of row a3 that can be got
// wrong silently — 2^67 * pixel_index steps cannot be checked by stepping.
#include <rocrand/rocrand_xorwow.h>

namespace {
// the engine keeps its state protected; a derived class may read and write it (no library code is altered)
struct RawEngine : rocrand_device::xorwow_engine {
    RawEngine(const unsigned int* x, unsigned int d) : rocrand_device::xorwow_engine(0ULL, 0ULL, 0ULL) {
        for (int k = 0; k < 5; ++k) m_state.x[k] = x[k];
        m_state.d = d;
    }
    void store(unsigned int* x, unsigned int* d) const {
        for (int k = 0; k < 5; ++k) x[k] = m_state.x[k];
        *d = m_state.d;
    }
};
}   // namespace

// n raw states (x: n x 5 words, d: n words), each advanced by subsequence[i] * 2^67 steps, then offset[i] steps, in place
extern "C" int ref_xorwow_skip(long long n, unsigned int* x, unsigned int* d, const unsigned long long* subsequence,
                               const unsigned long long* offset) {
    for (long long i = 0; i < n; ++i) {
        RawEngine e(x + 5 * i, d[i]);
        e.discard_subsequence(subsequence[i]);   // rocrand_xorwow.h: jump(subsequence, h_xorwow_sequence_jump_matrices)
        e.discard(offset[i]);                    // jump(offset, h_xorwow_jump_matrices); d += offset * 362437
        e.store(x + 5 * i, d + i);
    }
    return 0;
}

// one raw state stepped `count` times: out[j] = j-th 32-bit output; the state is updated in place
extern "C" int ref_xorwow_next(unsigned int* x, unsigned int* d, long long count, unsigned int* out) {
    RawEngine e(x, *d);
    for (long long j = 0; j < count; ++j) out[j] = e.next();
    e.store(x, d);
    return 0;
}
