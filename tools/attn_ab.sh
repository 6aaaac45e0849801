# build attention.hip with different -D settings into variants/ and time each with tools/attn_bench.py (run the timing on the GPU box)
#   bash tools/attn_ab.sh build "name1:-DX=1 -DY=2" "name2:..."      (CPU container)
#   bash tools/attn_ab.sh run [n_seq] [seqlen]                       (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift; mkdir -p variants; rm -f variants/*.so
  for spec in "$@"; do
    name="${spec%%:*}"; flags="${spec#*:}"
    ( cd spatialthinker_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $flags -c attention.hip -o /tmp/attn_$name.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^attention.o$') /tmp/attn_$name.o -o ../../variants/$name.so ) || exit 1
  done
else
  shift
  for so in variants/*.so; do echo "== $so"; ST_LIB=$so python3 tools/attn_bench.py ${1:-4} ${2:-1614} 2>&1 | grep -E "causal|full|err"; done
fi
