#!/bin/bash
# distance-factor tables on / off on the bench workload (JF_DTAB=0 switches them off in bench.py), one box, one call
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_dtab; mkdir -p $OUT
for rep in 1 2; do for v in 1 0; do
  JF_DTAB=$v python3 bench.py --no-pmc --steps 512 --cpu-sample-blocks 132 > $OUT/dtab${v}_$rep.json 2> $OUT/dtab${v}_$rep.err
  python3 -c "
import json; d=json.loads(open('$OUT/dtab${v}_$rep.json').read().strip().splitlines()[-1]); print('tables $v rep $rep: value %.4e step %.4f launch %.4f verified %s share %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['verified'], d['config'].get('distance_tables')))"
done; done
JF_DTAB=1 python3 bench.py --steps 512 --cpu-sample-blocks 132 > $OUT/dtab1_pmc.json 2> $OUT/dtab1_pmc.err
python3 -c "
import json; d=json.loads(open('$OUT/dtab1_pmc.json').read().strip().splitlines()[-1]); r=d['roofline']; print('tables 1 pmc: value %.4e launch %.4f valu/sb %.0f hbm %.3f GB frac_pmc %.3f' % (d['value'], r['avg_launch_ms'], r['issue']['valu_insts_per_source_block'], r['hbm']['bytes_per_launch']/1e9, r['frac_pmc']))"
