# Timing builds of the x3 tile Linear (results meaningless): average node_layer_x3_kernel / node_linear time in the 2D chain
set -e
cd $GRAFT_REPO_ROOT
C=semantically_consistent_graph_generation_amd/csrc
run() {
  rm -rf /tmp/abl_$1; mkdir -p /tmp/abl_$1
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$1 -o c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --T 10 --no-cpu-baseline > /tmp/abl_$1/log.txt 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/abl_$1/**/c_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'node_layer_x3' in r['Name'] or 'node_linear_x3' in r['Name']: print('$1', r['Name'][12:44], r['AverageNs'])
PY
}
run base
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA=-DSCGG_ABL_TILE_NO_MFMA > gpurun_out/make1.log 2>&1; run no_mfma
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA=-DSCGG_ABL_TILE_NO_W > gpurun_out/make2.log 2>&1; run no_w
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA="-DSCGG_ABL_TILE_NO_W -DSCGG_ABL_TILE_NO_MFMA" > gpurun_out/make3.log 2>&1; run no_w_no_mfma
