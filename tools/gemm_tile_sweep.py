"""Forced conv_gemm tile configurations (QA_GEMM_CFG) against the cost model choice on the LINEAR shapes of the codec graphs: TFLOP/s, back to back on resident operands.\nusage: python tools/gemm_tile_sweep.py   (the first row runs before the clocks have settled)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
import unified_audio_amd as qa
from unified_audio_amd import _lib
dev=torch.device("cuda:0"); lib=qa.load_library(); st=torch.cuda.current_stream().cuda_stream
def timed(fn,reps=30):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); best=1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize(); best=min(best,e0.elapsed_time(e1)*1e3/reps)
    return best
print("M x N x K            auto    128x64  128x128  64x128  64x64   (TFLOP/s)")
for M,N,K in ((9056,512,512),(9056,512,2048),(9056,1536,512),(9056,2048,512),(8000,1024,1024),(8000,3072,1024),(8000,2048,1024),(8000,1024,2048),(16000,1024,3072),(16000,2304,1024),(16000,1024,2304),(4000,768,768),(4000,3072,768),(4000,768,3072)):
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)/K**0.5; y=torch.empty(M,N,device=dev)
    a=_lib.qa_conv_args(); a.x,a.w,a.y=x.data_ptr(),w.data_ptr(),y.data_ptr()
    a.B,a.T_in,a.C_in,a.T_out,a.N=1,M,K,M,N; a.ldx,a.ldy,a.ldr,a.ldg=K,N,N,N; a.ksize,a.stride=1,1
    row=[]
    for cfg in (-1,1,2,3,4):
        _lib.set_knob("QA_GEMM_CFG",cfg)
        t=timed(lambda:_lib.check(lib.qa_conv1d_cl(C.byref(a),st)))
        row.append(2.0*M*N*K/t/1e6)
    _lib.set_knob("QA_GEMM_CFG",-1)
    print(f"{M:6d} x {N:5d} x {K:5d}   "+"  ".join(f"{v:6.1f}" for v in row), flush=True)
