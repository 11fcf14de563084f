for v in "$@"; do
  echo "== $v"
  QA_LIBRARY=$PWD/tools/_variants/$v/libquarkaudio_hip.so QA_BENCH_ONLY=${SHAPES:-cal,mimi,bt,dec.k3,dec.w2,convnext.pw1} timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
