# r06 call 9 (run three times: first list, refined list with three alternating passes, block-pitch sweep): store-pattern probe
set -u
OUT=gpurun_out/r06_09; mkdir -p $OUT
timeout 600 tools/probes/bin/write_burst.bin | tee $OUT/write_burst.txt
timeout 600 tools/probes/bin/write_burst.bin pitch | tee $OUT/write_pitch.txt
timeout 600 tools/probes/bin/write_burst.bin xcd | tee $OUT/write_xcd.txt
