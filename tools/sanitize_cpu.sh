#!/bin/bash
# AddressSanitizer + UBSan on the CPU-side code (GPU sanitizers are not available on the pool): the C++ host layer's CPU test
# and the oracle's C files driven through oracle/capi.py.  tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer tests/cpp/test_prl_host.cpp prlib_amd/csrc/prl/prl_host.cpp \
    -Lprlib_amd -lprlib_hip -Loracle -lprl_oracle -Wl,-rpath,$ROOT/prlib_amd -Wl,-rpath,$ROOT/oracle -Wl,-rpath,/opt/rocm/lib -o /tmp/test_prl_host_asan
ASAN_OPTIONS=detect_leaks=0 /tmp/test_prl_host_asan cpu
(cd oracle && gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -D_DEFAULT_SOURCE -shared -fPIC prl_oracle*.c -lm -o /tmp/libprl_oracle_asan.so)
cat > /tmp/asan_oracle.py <<'PY'
import sys, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np
orig = ctypes.CDLL
def patched(path, *a, **k):
    return orig('/tmp/libprl_oracle_asan.so' if 'libprl_oracle' in str(path) else path, *a, **k)
ctypes.CDLL = patched
import oracle.capi as oc
from prlib_amd import synth
for (h, w) in [(40, 50), (97, 131), (1, 1), (33, 200)]:
    g = synth.page_numpy(max(h, 8), max(w, 8), 3)[:h, :w].copy()
    for m, win in [(oc.SAUVOLA, 15), (oc.WOLFJOLION, 9), (oc.FENG, 21), (oc.NICK, 31)]:
        if min(h, w) > win: oc.binarize(g, oc.make_params(m, win, 0.2, 1))
    col = np.repeat(g[..., None], 3, 2).copy()
    oc.bgnorm(g); oc.bgnorm(col)
    oc.thin((g < 128).astype(np.uint8) * 255, 0); oc.thin((g < 128).astype(np.uint8) * 255, 1)
    oc.rotate(col, 12.5); oc.rotate(g, 90.0)
    if h > 30 and w > 40:
        oc.deskew(synth.text_page_numpy(h + 80, w + 100, 2, skew_deg=2.0))
        oc.denoise(col[:24, :24].copy(), 10.0, threads=2)
        oc.binarize_lv_nofilters(col, 0.125, 10); oc.binarize_lv(col, 0.125, 25, 2.0)
print("oracle under ASan/UBSan: ok")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 /tmp/asan_oracle.py "$ROOT"
# the copy-thread pool of the host-list entries (prlib_amd/csrc/prl/work_pool.h) under ThreadSanitizer
g++ -O1 -g -std=c++17 -fsanitize=thread -pthread "$ROOT/tests/cpp/test_work_pool.cpp" -o /tmp/test_work_pool_tsan
/tmp/test_work_pool_tsan 2>&1 | tee /tmp/test_work_pool_tsan.log | tail -2
if grep -q "ThreadSanitizer" /tmp/test_work_pool_tsan.log; then echo "work_pool under TSan: FAILED"; exit 1; else echo "work_pool under TSan: ok"; fi
