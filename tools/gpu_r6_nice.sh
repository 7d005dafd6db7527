#!/bin/bash
# round 6: four samples taking turns on one device with the decoder's worker threads at lower priority (HLALA_BAM_NICE: the threads that feed the GPU stay ahead of them)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for nice in 0 10 19 0 10; do
  echo "HLALA_BAM_NICE=$nice"
  HLALA_BAM_NICE=$nice timeout 1500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --long-reads 0 --no-extras-but-e2e --resident-steps 0 --e2e-threads 0 --e2e-samples "4" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
e=json.loads(sys.stdin.read()).get('end_to_end', {})
print('  one sample: %s pairs/s (decode %s s, alignment and typing %s s)' % (e.get('value'), e.get('decode_s'), e.get('alignment_and_typing_s')))
for s in e.get('several_samples', []): print('  ', s.get('samples'), 'samples: %.0f pairs/s, wall %.2f s' % (s.get('value', 0), s.get('wall_s', 0)), [l[l.find('(BAM decode'):l.find(', of which')] for l in s.get('per_sample_lines', [])])
"
done
