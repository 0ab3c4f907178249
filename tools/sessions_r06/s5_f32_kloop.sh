# Round 6, second session, call 5: the one-product K loop with four-step stages + hsum through LDS: statistics unchanged
# (digests against the library built before the hsum change), float32 tests, the pieces probe, rocprofv3 of the workload.
set -u
R=$PWD; O=$R/gpurun_out/r06s5; mkdir -p $O
export TMPDIR=/tmp
python tools/probes/f32_map_digest.py > $O/digest_new.txt 2>&1
MTM_LIB_PATH=$R/ab_builds/pre_hsum/libmtm_hip.so python tools/probes/f32_map_digest.py > $O/digest_pre_hsum.txt 2>&1
if diff $O/digest_new.txt $O/digest_pre_hsum.txt > $O/digest_diff.txt; then echo "float32 statistics: digests identical ($(wc -l < $O/digest_new.txt) lines)"; else echo "DIGESTS DIFFER"; head -20 $O/digest_diff.txt; fi
timeout 900 python -m pytest tests -m gpu -q -x -k "float32 or f32 or uint16" -s > $O/pytest_f32.log 2>&1; grep -E "passed|failed|error|float32 bound" $O/pytest_f32.log | tail -5
grep -E "^E " $O/pytest_f32.log | head -20
timeout 900 python tools/probes/f32_pieces_probe.py quick > $O/f32_pieces_probe.txt 2>&1; grep -E "cfg3_32|DIFFERENT" $O/f32_pieces_probe.txt
bash tools/profile_workloads.sh r06s5 f32_4k32 > $O/profile_f32.log 2>&1; tail -8 $O/profile_f32.log | cut -c1-220
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc_wait -o wait -- python $R/tools/probes/workload.py f32_4k32 3 > $O/pmc_wait.log 2>&1
db=$(find $O/pmc_wait -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_pmc.py "$db" > $O/wl_f32_4k32_pmc_wait.csv; rm -rf $O/pmc_wait
cd $R
grep -h "bf16\|hsum" $O/wl_f32_4k32_pmc_sq.csv $O/wl_f32_4k32_pmc_wait.csv | cut -c1-120
