#!/bin/bash
# Calibration of the FETCH_SIZE reading and of the VALU issue peak (tools/calib.hip), each under its own pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r3}
O=$R/gpurun_out/$ROUND/calib
mkdir -p $O
cd $R
C=$R/tools/calib
timeout -k 10 120 $C stream 4096 3 > $O/stream.txt 2>&1 || { cat $O/stream.txt; exit 1; }
timeout -k 10 120 $C gather64 4096 3 > $O/gather_hbm.txt 2>&1 || exit 1
timeout -k 10 120 $C gather64 128 6 noflush > $O/gather_mall.txt 2>&1 || exit 1
# the rate against the table size: HBM-resident tables up to 2 GiB gather at 34 - 36 G records/s, from 4 GiB on at 19 (a cliff, not a slope)
for m in 512 1024 2048 4096 8192 16384; do timeout -k 10 120 $C gather64 $m 3 | tail -1; done > $O/gather_sizes.txt 2>&1 || exit 1
for k in gather128 gather256; do timeout -k 10 120 $C $k 4096 3 > $O/${k}_hbm.txt 2>&1 || exit 1; timeout -k 10 120 $C $k 128 6 noflush > $O/${k}_mall.txt 2>&1 || exit 1; done
timeout -k 10 120 $C valu 0 3 > $O/valu.txt 2>&1 || exit 1
timeout -k 10 120 $C valu_pk 0 3 > $O/valu_pk.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/p_stream -o p --output-format csv -- $C stream 4096 2 > $O/p_stream.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/p_gather_hbm -o p --output-format csv -- $C gather64 4096 2 > $O/p_gather_hbm.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/p_gather_mall -o p --output-format csv -- $C gather64 128 4 noflush > $O/p_gather_mall.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $O/p_gather_req -o p --output-format csv -- $C gather64 4096 2 > $O/p_gather_req.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $O/p_valu -o p --output-format csv -- $C valu 0 2 > $O/p_valu.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU -d $O/p_valu_pk -o p --output-format csv -- $C valu_pk 0 2 > $O/p_valu_pk.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/p_valu_pk_clk -o p --output-format csv -- $C valu_pk 0 2 > $O/p_valu_pk_clk.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/p_valu_clk -o p --output-format csv -- $C valu 0 2 > $O/p_valu_clk.log 2>&1 || exit 1
for k in stream gather_hbm gather_mall gather_req; do python3 tools/pmc_to_json.py "k_" $O/p_$k > $O/p_$k.json; done
python3 tools/pmc_to_json.py "k_valu(" $O/p_valu $O/p_valu_clk > $O/p_valu.json
python3 tools/pmc_to_json.py "k_valu_pk" $O/p_valu_pk $O/p_valu_pk_clk > $O/p_valu_pk.json
# the vector-memory pipe of a CU and the gather rate of dependent chains (tools/coop.hip): tables in L1, L2, the Infinity Cache, HBM
if [ -x $R/tools/coop ]; then for k in 16 2048 65536 1048576 2097152; do timeout -k 10 120 $R/tools/coop $k 1000 || exit 1; done > $O/coop.txt 2>&1; fi
cat $O/*.txt
echo "calib done"
