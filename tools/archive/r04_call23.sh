#!/bin/bash
# Is the 9 % of non-algorithmic HBM traffic of the dW launch the kernel's or the counter's?  FETCH_SIZE of the stand-alone read skeleton
# (tools/probes/probe_read_pattern.bin: 9 x 8192 x 32 KiB = 2,415,919,104 B per launch, nothing written, hipMalloc'd buffers)
OUT=$(pwd)/gpurun_out/r04c23; mkdir -p $OUT
BIN=$(pwd)/tools/probes/probe_read_pattern.bin
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc -o p -- $BIN > $OUT/probe.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
fs=glob.glob(out+'/pmc/**/*counter_collection.csv', recursive=True)
csv.field_size_limit(1<<30)
per=collections.defaultdict(float)
for row in csv.DictReader(open(fs[0])):
    if row['Counter_Name']=='FETCH_SIZE': per[(row['Kernel_Name'][:40], row['Dispatch_Id'])]+=float(row['Counter_Value'])
v=sorted(per.values())
alg=9*8192*32768
print('launches %d  FETCH_SIZE KB: min %.0f median %.0f max %.0f   2 x FETCH_SIZE x 1024 / algorithmic bytes: min %.4f median %.4f max %.4f'
      % (len(v), v[0], v[len(v)//2], v[-1], 2*v[0]*1024/alg, 2*v[len(v)//2]*1024/alg, 2*v[-1]*1024/alg))
PY
find $OUT/pmc -name "*.csv" -size +1M -delete
