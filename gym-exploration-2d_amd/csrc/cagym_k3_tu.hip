// cagym_k3_tu.hip -- one translation unit per generation-3 specialisation: compiled with
//   -DK3_NT=<lanes> -DK3_MT=<compile-time M> -DK3_WP=<worlds per workgroup> -DK3_OBST=<0|1>
// (gym-exploration-2d_amd/build.py), or included by cagym_api.hip under -DCAGYM_MONOLITHIC.
#include <hip/hip_runtime.h>
#ifndef CAGYM_MONOLITHIC
#define CAGYM_K3_UNIT 1  // cagym_kernels.h: device functions only, its __global__ kernels belong to cagym_api.hip
#endif
#include "cagym_kernels3.h"
#include "cagym_split3.h"
#include "cagym_launch3.h"

#if !defined(K3_NT) || !defined(K3_MT) || !defined(K3_WP) || !defined(K3_OBST)
#error "cagym_k3_tu.hip needs K3_NT, K3_MT, K3_WP and K3_OBST"
#endif
#define K3_CAT_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define K3_CAT(a, b, c, d, e) K3_CAT_(a, b, c, d, e)

void K3_CAT(cagym_k3_launch_, K3_NT, K3_MT, K3_WP, K3_OBST)(const K3Launch& L) {
    constexpr bool OB = K3_OBST != 0;
    const dim3 g(L.grid), b(K3_NT);
    if (L.half == K3_HALF_PRE) {  // the split step (cagym_split3.h): L.lds is the half's own footprint
        hipLaunchKernelGGL((k_step_pre3<K3_NT, K3_MT, K3_WP, OB>), g, b, L.lds, L.stream, L.D);
    } else if (L.half == K3_HALF_POST) {
        if (L.auto_reset) hipLaunchKernelGGL((k_step_post3<K3_NT, K3_MT, K3_WP, true, OB>), g, b, L.lds, L.stream, L.D, L.ext, L.out);
        else hipLaunchKernelGGL((k_step_post3<K3_NT, K3_MT, K3_WP, false, OB>), g, b, L.lds, L.stream, L.D, L.ext, L.out);
    } else if (L.rollout) {
        if (L.auto_reset) hipLaunchKernelGGL((k_rollout3<K3_NT, K3_MT, K3_WP, true, OB>), g, b, L.lds, L.stream, L.D, L.n_steps, L.out, L.any_rvo);
        else hipLaunchKernelGGL((k_rollout3<K3_NT, K3_MT, K3_WP, false, OB>), g, b, L.lds, L.stream, L.D, L.n_steps, L.out, L.any_rvo);
    } else {
        if (L.auto_reset) hipLaunchKernelGGL((k_step3<K3_NT, K3_MT, K3_WP, true, OB>), g, b, L.lds, L.stream, L.D, L.ext, L.out, L.any_rvo);
        else hipLaunchKernelGGL((k_step3<K3_NT, K3_MT, K3_WP, false, OB>), g, b, L.lds, L.stream, L.D, L.ext, L.out, L.any_rvo);
    }
}

// > 64 KiB of dynamic LDS needs the attribute raised
void K3_CAT(cagym_k3_setattr_, K3_NT, K3_MT, K3_WP, K3_OBST)(int lds) {
    constexpr bool OB = K3_OBST != 0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step3<K3_NT, K3_MT, K3_WP, false, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step3<K3_NT, K3_MT, K3_WP, true, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout3<K3_NT, K3_MT, K3_WP, false, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout3<K3_NT, K3_MT, K3_WP, true, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    // the two halves of a split step never need more than the fused kernel
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_pre3<K3_NT, K3_MT, K3_WP, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_post3<K3_NT, K3_MT, K3_WP, false, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_post3<K3_NT, K3_MT, K3_WP, true, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}
#undef K3_CAT
#undef K3_CAT_
