for m in 0 64 256 320 130 131; do ABL_MASK=$m timeout -k 10 120 python tools/bench_afterstates.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant $m', 'valid-only %.3f ms' % d['include_terminal=False']['ms'], ' with-all %.3f ms' % d['include_terminal=True']['ms'])
" || exit 1; done
