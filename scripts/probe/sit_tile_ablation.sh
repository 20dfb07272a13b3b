set -e
cd $GRAFT_REPO_ROOT
C=semantically_consistent_graph_generation_amd/csrc
run() { timeout -k 10 300 python scripts/bench_sit.py 2>/dev/null | python -c "import sys,json; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_ancestral_step'])"; }
run base
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA=-DSCGG_ABL_TILE_NO_MFMA > gpurun_out/make1.log 2>&1; run no_mfma
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA=-DSCGG_ABL_TILE_NO_W > gpurun_out/make2.log 2>&1; run no_w
touch $C/sit_block.hip $C/node_layer.hip; make -C $C -j16 EXTRA="-DSCGG_ABL_TILE_NO_W -DSCGG_ABL_TILE_NO_MFMA" > gpurun_out/make3.log 2>&1; run no_w_no_mfma
