#!/bin/bash
# Build libqampy_hip.so for gfx950 (cross-compiles without a GPU).  Usage: qampy_amd/csrc/build.sh [extra hipcc flags]
# One object per translation unit, rebuilt when its source or any header is newer; $(nproc) compilers at a time.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
mkdir -p build
UNITS="api train_f32 train_f64 apply bps ser synth"
for m in mrde cma rde mcma sbd mddma dd cma2; do UNITS="$UNITS train_seg_${m}_f32 train_seg_${m}_f64"; done
for m in cma mcma mddma sbd; do UNITS="$UNITS train_seg_${m}_f32_ad"; done
todo=""
for f in $UNITS; do
    stale=0
    [ -f build/$f.o ] || stale=1
    # dependencies: the headers the compiler reported for this unit last time (build/<unit>.d, -MMD), else every header
    deps="$f.hip common.h train_impl.h train_la.h train_bi.h train_pit.h train_seg.h ../../include/qampy_hip.h"
    if [ -f build/$f.d ]; then
        deps="$f.hip $(tr -d '\\\n' < build/$f.d | sed 's/^[^:]*://' | tr ' ' '\n' | grep -v '^/opt/' | grep -v '^/usr/' | grep -v '^$' | sort -u | tr '\n' ' ')"
    fi
    for dep in $deps; do
        [ $stale = 1 ] || { [ -e $dep ] && [ $dep -nt build/$f.o ] && stale=1; } || true
    done
    [ $stale = 1 ] && todo="$todo $f"
done
if [ -n "$todo" ]; then
    echo $todo | tr ' ' '\n' | xargs -P ${QH_BUILD_JOBS:-$(nproc)} -I{} $HIPCC $FLAGS -MMD -MF build/{}.d -c {}.hip -o build/{}.o
fi
OBJS=""
for f in $UNITS; do OBJS="$OBJS build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libqampy_hip.so $OBJS
echo "built $(cd .. && pwd)/libqampy_hip.so"
