# Build variants/trace512.so (the library with gemm_swiglu512.hip compiled -DST_GU512_TRACE; variants/ is git-ignored) — run in the build container:
#   bash tools/gu512_phase_trace.sh       then on the GPU box:  ST_LIB=variants/trace512.so python tools/gu512_phase_trace.py 512
set -e
cd "$(dirname "$0")/../spatialthinker_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DST_GU512_TRACE -c gemm_swiglu512.hip -o /tmp/gemm_swiglu512_trace.o
objs=$(ls *.o | grep -v '^gemm_swiglu512.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_swiglu512_trace.o -o ../../variants/trace512.so
ls -la ../../variants/trace512.so
