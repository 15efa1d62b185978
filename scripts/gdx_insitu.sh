# In-situ A/B of the fused gdMlp kernel's variants: the eval bench's own HIP-event timing of the level-0 / level-1 launches and its img/s
# (BEM_GDX_VARIANT bit 0: packed phase B, bit 1: halo blocks loaded one after the other).
for v in 0 1 2 3; do
  for key in "gdmlp_x6<3>" "gdmlp_x6<5>"; do
    BEM_GDX_VARIANT=$v python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --profile-kernel "$key" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('variant $v  %-12s avg launch %7.1f us   %6.1f img/s' % ('$key', d['roofline']['avg_launch_us'], d['value']))"
  done
done
