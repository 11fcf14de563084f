// mfma4_layout.hip - prints the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 on the device it runs on
// (the narrow-tile GEMV of csrc/lm_decode.hip assumes: A lane l -> block l>>2, row l&3; B lane l -> block l>>2, col l&3;
// D lane l, VGPR r -> block l>>2, row r, col l&3).  Build: hipcc --offload-arch=gfx950 -O2 tools/micro/mfma4_layout.hip -o mfma4_layout
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    // A[block][row] = 1 + row + 10 * block, B[block][col] = 100 * (1 + col): D[block][row][col] = A * B identifies (block, row, col)
    const float a = 1.f + (lane & 3) + 10.f * (lane >> 2);
    const float b = 100.f * (1 + (lane & 3));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    float* d;
    float h[256];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const float want = (1.f + r + 10.f * (lane >> 2)) * 100.f * (1 + (lane & 3));
            if (h[lane * 4 + r] != want) {
                if (bad < 8) printf("lane %d vgpr %d: got %.0f, assumed map gives %.0f\n", lane, r, h[lane * 4 + r], want);
                ++bad;
            }
        }
    printf(bad ? "MFMA4 LAYOUT MISMATCH (%d)\n" : "mfma_f32_4x4x1 layout as assumed (%d mismatches)\n", bad);
    return bad != 0;
}
