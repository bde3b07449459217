export TMPDIR=/tmp
for spec in "cfg4 49152" "cfg4 55296" "cfg4 61440" "cfg4 62500" "cfg4 73728" "cfg3 98304" "cfg3 100000" "cfg3 106496" "cfg3 114688"; do set -- $spec
timeout 600 python bench.py --config $1 --families $2 --steps 60 --warmup 5 --no-cpu-baseline --no-search --no-tables --no-strong --no-probes 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 F=$2: walk %.4f ms frac %.3f  grid cfg %s  per-family ns %.2f' % (r['avg_launch_ms'], r['frac'], d['engine'].split('cfg(nftw,nrtw,wf,wr)=')[1].split(' park')[0], 1e6*r['avg_launch_ms']/$2))"
done
