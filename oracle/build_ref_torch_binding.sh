#!/bin/bash
# Builds the reference's OWN PyTorch binding — cuTENSOR/python/cutensor/torch/einsum.cc (pybind11) over
# cuTENSOR/python/einsum.h — UNMODIFIED, from where the sources lie under /root/reference, against OUR
# include/cutensor.h + libcutensor.so, and stages the reference's python package around it so the reference's
# own test file (cutensor/torch/einsum_test.py) can run on the GPU box, where /root/reference does not exist.
#
# Everything lands under oracle/_ref/pyref/ (git-ignored: never in history; not gpurun-ignored: it travels with
# the snapshot like the built .so files).  Nothing under cudalibrarysamples_amd/ imports it: it is the checker.
#
#   oracle/_ref/pyref/cutensor/torch/binding*.so     einsum.cc compiled with hipcc, no hipify pass, no edits
#   oracle/_ref/pyref/cutensor/{__init__,common,package_info}.py, torch/{__init__,einsum,einsum_test}.py
#                                                    staged verbatim at build time (cp from the reference tree)
#
# torch-ROCm spells the CUDA context/allocator headers einsum.cc includes under ATen/hip + c10/hip;
# tests/sample_compat/torch_shim/ (fixture) maps those three names, tests/sample_compat/ maps <cuda_runtime.h>
# etc. exactly as for the .cu samples.  The reference's setup.py (needs nvcc + the closed libcutensor) is not used.
set -eu
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF=/root/reference/cuTENSOR/python
OUT="$ROOT/oracle/_ref/pyref"
[ -d "$REF/cutensor/torch" ] || { echo "reference tree not present: skipping"; exit 0; }
PY=${PYTHON:-python3}
mkdir -p "$OUT/cutensor/torch"
for f in __init__.py common.py package_info.py torch/__init__.py torch/einsum.py torch/einsum_test.py; do
    install -m 644 "$REF/cutensor/$f" "$OUT/cutensor/$f"
done
SUFFIX=$($PY -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
SO="$OUT/cutensor/torch/binding$SUFFIX"
if [ -f "$SO" ] && [ -z "$(find "$REF/cutensor/torch/einsum.cc" "$REF/einsum.h" "$ROOT/include" "$ROOT/tests/sample_compat" "${BASH_SOURCE[0]}" -newer "$SO" -type f -print -quit)" ]; then
    exit 0   # up to date (the module links libcutensor.so dynamically: a rebuilt library needs no relink)
fi
TORCH_DIR=$($PY -c 'import torch, os; print(os.path.dirname(torch.__file__))')
PYINC=$($PY -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
PB11=$($PY -c 'import pybind11; print(pybind11.get_include())' 2>/dev/null || echo "$TORCH_DIR/include")
ABI=$($PY -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
# same macros the reference's CustomExtension.Torch passes (c_extensions_utils.py:58-64)
hipcc -x c++ -std=c++17 -O2 -w -fPIC -shared -fopenmp \
    -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_API_INCLUDE_EXTENSION_H -DTORCH_EXTENSION_NAME=binding \
    -D_GLIBCXX_USE_CXX11_ABI=$ABI \
    -I"$ROOT/tests/sample_compat/torch_shim" -I"$ROOT/tests/sample_compat" -I"$ROOT/include" \
    -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I"$PYINC" -I"$PB11" -I/opt/rocm/include \
    "$REF/cutensor/torch/einsum.cc" -o "$OUT/cutensor/torch/binding$SUFFIX" \
    -L"$ROOT/cudalibrarysamples_amd/lib" -Wl,-rpath,'$ORIGIN/../../../../../cudalibrarysamples_amd/lib' -lcutensor \
    -L"$TORCH_DIR/lib" -Wl,-rpath,"$TORCH_DIR/lib" -lc10 -lc10_hip -ltorch_cpu -ltorch_hip -ltorch -ltorch_python \
    -L/opt/rocm/lib -lamdhip64 2> "$OUT/build.log" \
  && echo "built oracle/_ref/pyref/cutensor/torch/binding$SUFFIX" \
  || { echo "FAILED reference torch binding (see oracle/_ref/pyref/build.log)"; exit 1; }
