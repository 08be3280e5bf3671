#!/bin/bash
# TEST-HARNESS FIXTURE: starts N ranks of a program built against sample_compat/mpi.h on this node.
#   tests/sample_compat/mpirun.sh <N> <program> [args...]
# Rank r gets RANK=r WORLD_SIZE=N and a fresh CTAMD_MPI_DIR; the sample picks GPU (local rank % device count)
# itself (cutensorMp_contraction.cu:78-91), so N must not exceed the number of GPUs (RCCL: one rank per device).
set -u
N="$1"; shift
DIR="$(mktemp -d /tmp/ctamd_mpi.XXXXXX)"
pids=()
for ((r = 0; r < N; ++r)); do
    RANK=$r WORLD_SIZE=$N CTAMD_MPI_DIR="$DIR" "$@" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
rm -rf "$DIR"
exit $rc
