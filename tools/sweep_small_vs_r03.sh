# Development aid: device time of the search over a grid of shapes, this build against round 3's library (tools/exp/libslamhip_r03.so, built
# from the round-3 commit), in one process, bit-exactness checked; flags every shape that got slower by more than 1 %.
L=slam-experiments_amd/lib/libslamhip.so
SH=""
for n in 200 1000 4096 8192 20000 65536 262144 1048576; do for m in 200 1000 2000 4096 8192 12000 16000; do SH="$SH ${n}x${m}"; done; done
timeout -k 10 1000 python tools/ab_time.py tools/exp/libslamhip_r03.so,$L $SH --rounds 3 --reps 40 --check > gpurun_out/r04/ab_sweep_small.log 2>&1
python3 - <<'PY'
import re
rows={}
for line in open('gpurun_out/r04/ab_sweep_small.log'):
    m=re.match(r"(\d+x\d+)\s+(\S+): median\s+([\d.]+) us.*first library's: (\w+)", line)
    if m: rows.setdefault(m.group(1),{})[m.group(2)]=(float(m.group(3)), m.group(4))
bad=[]
for k,v in rows.items():
    if '_r03' in v and 'shipped' in v:
        r=v['shipped'][0]/v['_r03'][0]
        flag = " <-- slower" if r>1.01 else ""
        if v['shipped'][1]!='True': flag+=" TABLE DIFFERS"
        print(f"{k:>16} r03 {v['_r03'][0]:10.1f}  now {v['shipped'][0]:10.1f}  ratio {r:.3f}{flag}")
PY
