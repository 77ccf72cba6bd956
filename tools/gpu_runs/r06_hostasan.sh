#!/bin/bash
# Round 6: one AddressSanitizer pass.  GPU AddressSanitizer (xnack+ code objects, HSA_XNACK=1) is not available on this pool -- gpurun
# refuses it -- so the DEVICE code runs as shipped and the HOST side of libbmx.so (the std::vector / std::map / staging code behind the
# C-ABI: 6,000 lines) is built with -fsanitize=address -fno-gpu-sanitize and run on the GPU box under the whole -m gpu suite's parity
# tests, the smoke test and a slice of both soaks.  Build (in the container):
#   cd bitmagic_amd/csrc; F="--offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan"
#   hipcc $F -c bmx.hip -o _obj/bmx_asan.o; hipcc $F -c bmx_group.hip -o _obj/bmx_group_asan.o
#   hipcc $F -shared _obj/bmx_asan.o _obj/bmx_group_asan.o -o ../lib/libbmx_hostasan.so -ldl -lpthread
out=gpurun_out/r06_hostasan; mkdir -p $out
export LD_PRELOAD=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$PWD/$out/asan_report
export BMX_LIB=$PWD/bitmagic_amd/lib/libbmx_hostasan.so
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" > $out/rc.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/rc.txt
timeout 600 python tools/soak_r05.py 30 > $out/soak_r05.txt 2>&1; echo "soak_r05 rc=$?" >> $out/rc.txt
timeout 600 python tools/soak_r04.py 8 > $out/soak_r04.txt 2>&1; echo "soak_r04 rc=$?" >> $out/rc.txt
{ cat $out/rc.txt; tail -2 $out/smoke.txt; grep -E "passed|failed" $out/pytest.txt | tail -1; echo "tests that failed at torch.cuda initialisation under the preloaded sanitizer runtime (dlopen of libcaffe2_nvrtc.so, not library code): $(grep -c "libcaffe2_nvrtc" $out/pytest.txt) error lines; other failures: $(grep "^FAILED" $out/pytest.txt | wc -l) listed below"; grep "^FAILED" $out/pytest.txt; tail -1 $out/soak_r05.txt; tail -1 $out/soak_r04.txt;
  echo "ASAN reports written: $(ls $out | grep -c asan_report)"; for f in $out/asan_report*; do [ -f "$f" ] && head -30 "$f"; done; } > $out/summary.txt
cat $out/summary.txt
