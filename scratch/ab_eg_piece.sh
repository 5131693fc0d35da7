#!/bin/bash
# token-segment sums by rows per piece (EG_PIECE): the embed_grad row of bench.py's per-kernel rooflines, three library builds in one session
R=$GRAFT_REPO_ROOT; cd /tmp
for rnd in 1 2; do for P in 256 512 1024; do
  cp $R/scratch/lib_p$P.so $R/music-fader-nets_amd/libfadernets_hip.so
  python $R/bench.py --steps 10 --warmup 3 --sustain 0 --no-x6 --no-cpu-baseline --no-decode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); e=d['roofline_all']['embed_grad']; print('EG_PIECE $P: step', d['ms_per_step'], 'ms; segment sums', e['avg_launch_us'], 'us per launch =', e['frac'], 'of the HBM peak')"
done; done
