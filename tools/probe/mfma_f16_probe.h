// TOOLS ONLY (tools/build_f16_probe.py): force-included in front of the kernel sources, it turns every v_mfma_f32_32x32x16_bf16 of the bf16
// kernels into v_mfma_f32_32x32x16_f16 ON THE SAME BITS -- numerically meaningless, a timing / clock probe: does the fp16 matrix
// instruction run these loops at the bf16 instruction's rate and clock?  (the question behind the fp16-piece split of DESIGN 10.)
#pragma once
typedef _Float16 lu_probe_half8 __attribute__((ext_vector_type(8)));
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(lu_probe_half8, a), __builtin_bit_cast(lu_probe_half8, b), c, x, y, z)
