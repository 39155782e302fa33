// fp4_probe.hip — what v_cvt_scalef32_pk_f16_fp4 does with a dword (which byte, which nibble first) and v_dot2_f32_f16's sum:  hipcc --offload-arch=gfx950 -O2 tools/fp4_probe.hip -o tools/fp4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned * in, float * out) {
    const unsigned v = in[0];
    h2 a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(v, 1.0f, 0), a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(v, 1.0f, 1);
    h2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(v, 1.0f, 2), a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(v, 1.0f, 3);
    out[0] = (float) a0.x; out[1] = (float) a0.y; out[2] = (float) a1.x; out[3] = (float) a1.y;
    out[4] = (float) a2.x; out[5] = (float) a2.y; out[6] = (float) a3.x; out[7] = (float) a3.y;
    h2 b = { (_Float16) 3.0f, (_Float16) -100.0f };
    out[8] = __builtin_amdgcn_fdot2(a0, b, 0.25f, false);
    h2 s = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(v, 4.0f, 0);
    out[9] = (float) s.x; out[10] = (float) s.y;
}
int main() {
    unsigned * d_in; float * d_out; float h[16];
    hipMalloc(&d_in, 4); hipMalloc(&d_out, 64);
    const unsigned vals[3] = { 0x76543210u, 0xFEDCBA98u, 0x000000A3u };
    for (int t = 0; t < 3; t++) {
        hipMemcpy(d_in, &vals[t], 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d_in, d_out);
        hipMemcpy(h, d_out, 44, hipMemcpyDeviceToHost);
        printf("%08x:", vals[t]); for (int i = 0; i < 11; i++) printf(" %g", h[i]); printf("\n");
    }
    return 0;
}
