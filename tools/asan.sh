#!/bin/bash
# CPU-side sanitizer pass (build container only — never on the GPU box: GPU AddressSanitizer is not available on this pool).
# A copy of the tree is built with -fsanitize=address,undefined for everything that runs on the HOST without a GPU:
#   * oracle/ (the C restatement the parity tests trust),
#   * the pybind11 host module (host/metadynamics_host.cc, grid_file.h: grid-file parser, hills log, argument validation),
# and `pytest -m "not gpu"` runs inside the copy with the sanitizer runtime preloaded into python.  libmtd_hip.so itself (device
# code + HIP host glue) is the regular build.  usage: tools/asan.sh [logfile]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG="${1:-$ROOT/profiles/r4/asan_ubsan_cpu_tests.log}"
WORK="${TMPDIR:-/tmp}/mtd_asan_tree"
rm -rf "$WORK" && mkdir -p "$WORK"
tar -C "$ROOT" --exclude=.git --exclude=gpurun_out --exclude=__pycache__ --exclude=.pytest_cache --exclude='tools/bin' -cf - . | tar -C "$WORK" -xf -
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
make -C "$WORK/oracle" -s clean
make -C "$WORK/oracle" -s CFLAGS="-O1 -fPIC -std=gnu99 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off $SAN"
[ -d /root/reference/metadynamics ] && make -C "$WORK/oracle" -s _ref CXXFLAGS="-O1 -fPIC -std=c++14 -Wno-deprecated-declarations $SAN"
rm -f "$WORK"/metadynamics-plugin_amd/metadynamics/_metadynamics*.so
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
PBINC=$(python3 -c "import pybind11; print(pybind11.get_include())")
make -C "$WORK/metadynamics-plugin_amd/host" -s CXXFLAGS="-O1 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -I$WORK/include -isystem /opt/rocm/include -I$PYINC -I$PBINC $SAN"
mkdir -p "$(dirname "$LOG")"
{
echo "# tools/asan.sh: oracle/ and the host module built with: $SAN"
echo "# gcc $(gcc -dumpversion); $(date -u +%Y-%m-%dT%H:%MZ); pytest -m 'not gpu' inside $WORK with libasan preloaded"
cd "$WORK"
# detect_leaks=0: CPython and torch keep allocations for the life of the process; everything else is fatal
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1
echo "# exit status: $?"
} | tee "$LOG"
grep -q "# exit status: 0" "$LOG"
