#!/bin/bash
# fine-grained PLONK timeline (kernels >= 20 us) to see what fills the gaps between the rounds
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r3r; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inputs --no-2p24 --no-micro > $O/bench.json 2> $O/err.log
ls $O/tr/*/ | head
python3 $R/tools/timeline_proof.py $O/tr plonk 20 > $O/timeline_plonk_fine.txt 2>&1
F=$(ls $O/tr/*/*memory_copy_trace.csv 2>/dev/null | head -1); [ -n "$F" ] && gzip -c $F > $O/memcpy.csv.gz
K=$(ls $O/tr/*/*kernel_trace.csv | head -1); python3 - "$K" > $O/plonk_kernels.csv <<'PY'
import csv,sys
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Queue_Id"],r["Kernel_Name"][:60],r["Grid_Size_X"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# last plonk proof: from the second-to-last k_quotient to the last
q=[r[0] for r in rows if "k_quotient" in r[3]]
a,b=q[-2],q[-1]
for r in rows:
    if a<=r[0]<=b: print("%.3f,%.3f,%s,%s,%s"%((r[0]-a)/1e6,(r[1]-r[0])/1e6,r[2],r[3],r[4]))
PY
rm -rf $O/tr
wc -l $O/timeline_plonk_fine.txt $O/plonk_kernels.csv
