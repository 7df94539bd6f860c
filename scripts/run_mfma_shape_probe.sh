# probes/mfma_shape_probe with hwmon sampled beside it (all cards; the one whose power moves is the card under test)
mkdir -p gpurun_out/r5a
( while true; do for c in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "$(date +%s.%N) $c $(cat $c/power1_input 2>/dev/null || cat $c/power1_average) $(cat $c/freq1_input)"; done; sleep 0.05; done ) > /tmp/hw.txt &
S=$!
./probes/mfma_shape_probe > gpurun_out/r5a/mfma_shape.txt 2>&1
kill $S
python - >> gpurun_out/r5a/mfma_shape.txt <<'PY'
import collections
rows = [l.split() for l in open('/tmp/hw.txt') if len(l.split()) == 4]
phases = [l.split(None, 3) for l in open('gpurun_out/r5a/mfma_shape.txt') if l.startswith('PHASE')]
cards = sorted({r[1] for r in rows})
def med(v): v = sorted(v); return v[len(v) // 2] if v else float('nan')
span = {c: max(float(r[2]) for r in rows if r[1] == c) - min(float(r[2]) for r in rows if r[1] == c) for c in cards}
card = max(span, key=span.get)
print('card under test:', card)
for _, t0, t1, name in phases:
    t0, t1 = float(t0), float(t1)
    sel = [r for r in rows if r[1] == card and t0 + 0.4 < float(r[0]) < t1]
    print(f"{name.strip()[:46]:46s} power {med([float(r[2]) for r in sel]) / 1e6:6.0f} W   sclk {med([float(r[3]) for r in sel]) / 1e6:5.0f} MHz")
PY
cat gpurun_out/r5a/mfma_shape.txt
