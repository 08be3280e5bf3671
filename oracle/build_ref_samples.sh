#!/bin/bash
# Builds the reference's OWN sample drivers, from the sources where they lie under /root/reference,
# against OUR libcutensor.so / libcutensorMg.so / libcutensorMp.so.  Nothing is copied into the repository: outputs go to
# oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).  The sources are compiled
# unmodified with `hipcc -x hip`; tests/sample_compat/ supplies <cuda_runtime.h>/<cuda_fp16.h> (and, for the
# cutensorMp sample, <mpi.h>/<nccl.h>/<cuComplex.h>/<cuda_profiler_api.h>) for the names the samples call themselves.  The reference's build system (Makefile/CMake, needs nvcc
# and the closed libcutensor) is not used.
set -u
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF=/root/reference
OUT="$ROOT/oracle/_ref"
[ -d "$REF/cuTENSOR" ] || { echo "reference tree not present: skipping"; exit 0; }
mkdir -p "$OUT"
FLAGS="-x hip --offload-arch=gfx950 -std=c++17 -O2 -w -I$ROOT/tests/sample_compat -I$ROOT/include"
LINK="-L$ROOT/cudalibrarysamples_amd/lib -Wl,-rpath,\$ORIGIN/../../cudalibrarysamples_amd/lib"
# a target is rebuilt only when its source, the ABI headers or the fixture headers are newer than it (the binaries link
# the library dynamically: a rebuilt libcutensor.so needs no relink)
fresh() {   # fresh <output> <source>
    [ -x "$1" ] || return 1
    [ -z "$(find "$2" "$ROOT/include" "$ROOT/tests/sample_compat" "${BASH_SOURCE[0]}" -newer "$1" -type f -print -quit)" ]
}
rc=0
for s in contraction einsum reduction elementwise_permute elementwise_binary elementwise_trinary elementwise_permute_padding \
         contraction_plan_cache contraction_jit contraction_trinary blocksparse; do
    if fresh "$OUT/$s" "$REF/cuTENSOR/$s.cu"; then continue; fi
    if hipcc $FLAGS "$REF/cuTENSOR/$s.cu" -o "$OUT/$s" $LINK -lcutensor 2> "$OUT/$s.log"; then
        echo "built oracle/_ref/$s"
    else
        echo "FAILED oracle/_ref/$s (see oracle/_ref/$s.log)"; rc=1
    fi
done
if fresh "$OUT/contraction_multi_gpu" "$REF/cuTENSORMg/contraction_multi_gpu.cu"; then :
elif hipcc $FLAGS "$REF/cuTENSORMg/contraction_multi_gpu.cu" -o "$OUT/contraction_multi_gpu" $LINK -lcutensorMg -lcutensor 2> "$OUT/contraction_multi_gpu.log"; then
    echo "built oracle/_ref/contraction_multi_gpu"
else
    echo "FAILED oracle/_ref/contraction_multi_gpu (see oracle/_ref/contraction_multi_gpu.log)"; rc=1
fi
if fresh "$OUT/blog_post" "$REF/cuTENSORMg/blog_post.cu"; then :
elif hipcc $FLAGS "$REF/cuTENSORMg/blog_post.cu" -o "$OUT/blog_post" $LINK -lcutensorMg -lcutensor 2> "$OUT/blog_post.log"; then
    echo "built oracle/_ref/blog_post"
else
    echo "FAILED oracle/_ref/blog_post (see oracle/_ref/blog_post.log)"; rc=1
fi
if fresh "$OUT/cutensorMp_contraction" "$REF/cutensorMp/cutensorMp_contraction.cu"; then :
elif hipcc $FLAGS "$REF/cutensorMp/cutensorMp_contraction.cu" -o "$OUT/cutensorMp_contraction" $LINK -lcutensorMp -lcutensor -L/opt/rocm/lib -lrccl 2> "$OUT/cutensorMp_contraction.log"; then
    echo "built oracle/_ref/cutensorMp_contraction"
else
    echo "FAILED oracle/_ref/cutensorMp_contraction (see oracle/_ref/cutensorMp_contraction.log)"; rc=1
fi
exit $rc
