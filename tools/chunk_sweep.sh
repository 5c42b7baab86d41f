#!/bin/bash
# scan / prune time against the chunk size of the scan's dynamic strip assignment: bash tools/chunk_sweep.sh [sizes...]
for cs in ${@:-1 2 4 8}; do
  RSB_CHUNK_STRIPS=$cs timeout -k 10 120 python3 -u bench.py --steps 6 --warmup 1 --no-cpu-baseline --steps-in-flight 1 > gpurun_out/cs_$cs.json 2> gpurun_out/cs_$cs.err || exit 1
  python3 -c "
import json,sys;b=json.loads(open('gpurun_out/cs_$cs.json').read());s=b['stage_ms'];print('chunk_strips $cs: scan %.4f prune %.4f replay %.4f ms_per_step %.4f %s' % (s['scan'],s['prune'],s['device_replay'],b['ms_per_step'],b['config']['oracle_check']))"
done
