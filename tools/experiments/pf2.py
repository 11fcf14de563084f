"""conv_gemm K loop with a prefetch distance of one whole iteration (experiment for round 4; NOT in the product build).

Today the next chunk's global loads are issued in the first MFMA group of an iteration and waited for ~4 MFMAs (~250 cycles)
later, when the LDS stores of the same iteration need them (see the ISA of conv_gemm_kernel<128,128,2,2,false,16,true>:
global_load x4 around MFMA 14-16 of 32, s_waitcnt vmcnt(3) after MFMA 17) - an L2 hit takes longer than that and an HBM miss
four times as long, so every wave stalls once per chunk; with 2 waves per SIMD (launches of <= 512 tiles) nothing covers it.
Here the staging registers are loaded one iteration EARLIER: iteration kc first stores the registers (chunk kc + 1, loaded during
iteration kc - 1) into the free LDS buffer and then re-loads them with chunk kc + 2 - load -> use distance = 32 MFMAs (~2 000
cycles) wherever the compiler places the pair inside the group.  Same arithmetic, same LDS double buffer, same barriers:
bit-identical results.  Cost: the 4 staging float4 are live across the whole iteration (+ up to 16 VGPRs).

    python tools/variants.py --transform tools/experiments/pf2.py pf2 ""
    QA_LIBRARY=tools/_variants/pf2/libquarkaudio_hip.so python tools/lib_ab.py ...
"""


def transform(src: str) -> str:
    # only the BK = 16 instances (4 workgroups per CU at <= 128 VGPRs: 124 -> 128); at BK = 32 the longer live range would cost
    # the third wave per SIMD (166 -> 171 / 174 VGPRs), so those keep today's loop
    old = """    QA_LOAD_GLOBAL(0)
    QA_STORE_LDS(0)
    __syncthreads();
"""
    new = """    constexpr bool PF2 = BK == 16;
    QA_LOAD_GLOBAL(0)
    QA_STORE_LDS(0)
    if (PF2) {
        const int k1_ = min(1, nk - 1);
        QA_LOAD_GLOBAL(k1_)  // the staging registers now hold chunk 1
    }
    __syncthreads();
"""
    assert src.count(old) == 1
    src = src.replace(old, new)
    old = """            if (kk == QA_LOAD_AT) QA_LOAD_GLOBAL(nxt)
            if (kk == NKK - 1) QA_STORE_LDS(cur ^ 1)
"""
    new = """            if (PF2) {
                if (kk == 0) {  // registers = chunk kc + 1 (loaded one iteration ago) -> the free buffer; then chunk kc + 2 -> registers
                    QA_STORE_LDS(cur ^ 1)
                    const int nx2_ = min(kc + 2, nk - 1);
                    QA_LOAD_GLOBAL(nx2_)
                }
            } else {
                if (kk == QA_LOAD_AT) QA_LOAD_GLOBAL(nxt)
                if (kk == NKK - 1) QA_STORE_LDS(cur ^ 1)
            }
"""
    assert src.count(old) == 1
    return src.replace(old, new)


def transform_interleaved(src: str) -> str:
    """pf2 + the staging instructions spread under the wave's own MFMAs (one LDS store / one address add + one global load in the
    shadow of each of the first MFMAs of the group) instead of one clump after the second MFMA."""
    src = transform(src)
    old = """            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
"""
    new = """            if (PF2 && kk == 0) {
#pragma unroll
                for (int s_ = 0; s_ < A_IT + B_IT; ++s_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one LDS store of the staged chunk
                }
#pragma unroll
                for (int s_ = 0; s_ < A_IT + B_IT; ++s_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // the address add
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one global load of chunk kc + 2
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
"""
    assert src.count(old) == 1
    return src.replace(old, new)
