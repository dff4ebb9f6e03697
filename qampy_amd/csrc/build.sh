#!/bin/bash
# Build libqampy_hip.so for gfx950 (cross-compiles without a GPU).  Usage: qampy_amd/csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in api train_f32 train_f64 train_seg_a_f32 train_seg_b_f32 train_seg_a_f64 train_seg_b_f64 apply bps ser synth; do
    if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ train_impl.h -nt build/$f.o ] || [ train_la.h -nt build/$f.o ] || [ train_bi.h -nt build/$f.o ] || [ train_pit.h -nt build/$f.o ] || [ train_seg.h -nt build/$f.o ] || [ ../../include/qampy_hip.h -nt build/$f.o ]; then
        $HIPCC $FLAGS "$@" -c $f.hip -o build/$f.o &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libqampy_hip.so build/api.o build/train_f32.o build/train_f64.o build/train_seg_a_f32.o build/train_seg_b_f32.o build/train_seg_a_f64.o build/train_seg_b_f64.o build/apply.o build/bps.o build/ser.o build/synth.o
echo "built $(cd .. && pwd)/libqampy_hip.so"
