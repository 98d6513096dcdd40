#!/bin/bash
# Build an A/B copy of the library with extra -D flags on spam_spgemm.hip / spam_spmv.hip:
#   tools/build_variant.sh NAME "-DSPAM_WAVE_U=4"   ->  sparse_matrix_amd/libspam_hip_NAME.so  (use with PROBE_LIB=...)
set -e
cd "$(dirname "$0")/../sparse_matrix_amd/csrc"
NAME=$1; FLAGS=$2
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wall -Wno-unused-function"
make -s
for f in spam_spgemm spam_spmv; do /opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c $f.hip -o /tmp/${f}_$NAME.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libspam_hip_$NAME.so spam_core.o spam_scan.o /tmp/spam_spmv_$NAME.o spam_spmv_rg.o /tmp/spam_spgemm_$NAME.o spam_ewise.o
echo built ../libspam_hip_$NAME.so
