# L2-miss traffic and rate of the training GEMM tile vs the height of its tile groups.  variants/libst_hip_g{4,6,16}.so are builds of the
# library in which the three occurrences of the group height 8 in gemm_asm4.hip's tile map (`per_group = 8 * tiles_n`, `first_m = group * 8`,
# `gsz = min(tiles_m - first_m, 8)`, in the kernel and its two finish kernels) read 4 / 6 / 16 — e.g.
#   sed 's/per_group = 8 \* tiles_n/per_group = G * tiles_n/; s/first_m = group \* 8/first_m = group * G/; s/first_m, 8)/first_m, G)/' gemm_asm4.hip
# compiled with -DG=4 and linked with the other objects (the product source keeps the literal 8: its hash ties the committed PMC figures to it);
# the in-tree library = 8: rocprofv3 --kernel-trace --pmc FETCH_SIZE over tools/gemm_one.py on two shapes of the update pass.
#   bash tools/gemm_group_probe.sh      (from the repo root on the GPU box; results in gpurun_out/r06/gemm_group_probe.txt)
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out/r06
cp spatialthinker_amd/libst_hip.so /tmp/libst_hip_g8.so
out=gpurun_out/r06/gemm_group_probe.txt
: > $out
for g in 8 4 6 16; do
  if [ $g = 8 ]; then cp /tmp/libst_hip_g8.so spatialthinker_amd/libst_hip.so; else cp variants/libst_hip_g$g.so spatialthinker_amd/libst_hip.so; fi
  for shape in "20864 37888 3584" "20864 3584 18944"; do
    d=/tmp/ggp_${g}
    rm -rf $d
    (cd /tmp && PYTHONPATH=$root rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -o x -- python3 $root/tools/gemm_one.py 40 $shape > $d.log 2>&1) || tail -3 $d.log
    python3 - $g "$shape" $d >> $out <<'PY'
import csv, glob, sys
g, shape, d = sys.argv[1], sys.argv[2], sys.argv[3]
M, N, K = (int(x) for x in shape.split())
dur = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for f in glob.glob(d + '/**/x_kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f)) if 'gemm_nt4' in r['Kernel_Name']]
fetch = [float(r['Counter_Value']) for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True) for r in csv.DictReader(open(f)) if 'gemm_nt4' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
dur, fetch = sorted(dur)[len(dur) // 2], sorted(fetch)[len(fetch) // 2]
alg = (M * K + N * K + M * N) * 2
print(f"group rows {g:>2s}  M={M} N={N} K={K}: median {dur / 1e3:8.1f} us = {2 * M * N * K / dur / 1e3:7.1f} TF/s (under the counter pass), FETCH_SIZE x2 = {fetch * 2 * 1024 / 1e9:6.2f} GB vs {alg / 1e9:.2f} GB operands + result once = {fetch * 2 * 1024 / alg:.2f}x")
PY
  done
done
cp /tmp/libst_hip_g8.so spatialthinker_amd/libst_hip.so
cat $out
