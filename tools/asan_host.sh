#!/bin/bash
# AddressSanitizer on the HOST side of the library and of the C++ adapter (device code is not
# instrumented).  Step 1, anywhere:  tools/asan_host.sh build   -> build/asan/{libmg_hip.so,box_mg}
# Step 2, on the GPU box:           tools/asan_host.sh run     -> one line per driver case
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/asan
if [ "${1:-}" = build ]; then
  mkdir -p "$out" && cd "$root/multigrid_amd/csrc" &&
  /opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math \
      -Wno-unused-value -pthread -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
      mg_kernels.hip mg_sweep.hip mg_resrest.hip mg_ef.hip mg_api.hip mg_dist.hip -o "$out/libmg_hip.so" &&
  cd "$root/host" &&
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -o "$out/box_mg" \
      box_mg.cpp MultiGridSolverHIP.cpp PotentialSolver.cpp World.cpp Output.cpp -L"$out" -lmg_hip -Wl,-rpath,'$ORIGIN'
  exit $?
fi
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
B=$out/box_mg
n=0
while read -r args; do
  n=$((n + 1))
  timeout -k 10 120 $B $args > /tmp/asan_$n.log 2>&1
  rc=$?
  echo "box_mg $args: rc=$rc asan_errors=$(grep -c 'ERROR: AddressSanitizer' /tmp/asan_$n.log) $(grep -m1 converged= /tmp/asan_$n.log)"
  [ $rc -lt 124 ] || exit 1
done <<'CASES'
41 2 1e-6 --again
33 0 1e-6 --vti /tmp/asan.vti
65 3 1e-6 --pageable --dump /tmp/asan_phi.bin
65 3 1e-6 --b
129 3 1e-6 --gpus 0,0 --again
CASES
