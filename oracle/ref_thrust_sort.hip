// oracle/ref_thrust_sort.hip — TEST INFRASTRUCTURE, not part of the product.
//
// BVHNode's constructor sorts its span with `thrust::sort(l, l + n, BoxCompare(axis))` from inside ONE device thread
// (hitable/bvh.h:84-92; create_BVH<<<1,1>>>, createScene.h:42-44).  BoxCompare returns true on equal keys
// (bvh.h:36-41: `val1 - val2 < 0.0 ? false : true`) — not a strict weak ordering — so the order of a span with ties is
// a property of the ALGORITHM.  Without relocatable device code (CudaTest.vcxproj has no GenerateRelocatableDeviceCode
// / -rdc, so Thrust's CDP dispatch is compiled out) a device-side thrust::sort is Thrust's sequential backend,
// thrust/system/detail/sequential/stable_merge_sort.inl.  Thrust is not in the reference tree (CUDA 12.2 toolkit);
// this image ships the same library as rocThrust (/opt/rocm/include/thrust, same sequential backend sources).
//
// This file calls THAT code — nothing restated: thrust::sort with the sequential policy and BoxCompare's rule on a
// key array, on the host (ref_thrust_sort_host) and inside one device thread (ref_thrust_sort_device, what the
// reference does).  tests/test_ref_thrust.py holds the oracle's and the product's restatements (crt_oracle.c
// o_thrust_sort, crt_bvh.cpp thrust_sequential_sort) to its orders: committed fixtures
// tests/golden/thrust_sort_orders.npz (tests/make_thrust_fixtures.py) for machines without this library, the library
// itself where it exists, and the device run on the GPU box.
#include <hip/hip_runtime.h>
#include <thrust/sort.h>
#include <thrust/execution_policy.h>

namespace {

// bvh.h:9-45 with the GetBV lookups replaced by the key they read (box.min()[axis] of element a / b)
struct KeyCompare {
    const float* key;
    __host__ __device__ bool operator()(int a, int b) const {
        const float val1 = key[a], val2 = key[b];
        if (val1 - val2 < 0.0) return false;       // bvh.h:36
        else return true;                          // :39-41 (also for equal keys and for NaN)
    }
};

// mode 0: thrust::sort(thrust::seq, ...); mode 1: thrust::sort(...) without a policy, the reference's spelling
__global__ void sort_kernel(int mode, int n, const float* key, int* order) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    KeyCompare cmp{key};
    if (mode == 0) thrust::sort(thrust::seq, order, order + n, cmp);
    else thrust::sort(order, order + n, cmp);                                  // bvh.h:86: no policy
}

}  // namespace

extern "C" {

// order[0..n) in: a permutation of indices into key; out: sorted by the reference's comparator
int ref_thrust_sort_host(int n, const float* key, int* order) {
    if (n < 0) return -1;
    KeyCompare cmp{key};
    thrust::sort(thrust::seq, order, order + n, cmp);
    return 0;
}

int ref_thrust_sort_device(int mode, int n, const float* key, int* order) {
    if (n < 0 || mode < 0 || mode > 1) return -1;
    if (n == 0) return 0;
    float* dk = nullptr;
    int* dor = nullptr;
    if (hipMalloc((void**)&dk, n * sizeof(float)) != hipSuccess || hipMalloc((void**)&dor, n * sizeof(int)) != hipSuccess) return -2;
    int rc = 0;
    if (hipMemcpy(dk, key, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dor, order, n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = -2;
    if (rc == 0) {
        sort_kernel<<<1, 1>>>(mode, n, dk, dor);
        if (hipDeviceSynchronize() != hipSuccess) rc = -3;
        else if (hipMemcpy(order, dor, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = -4;
    }
    (void)hipFree(dk); (void)hipFree(dor);
    return rc;
}

}  // extern "C"
