cd "$(dirname "$0")/.."
for r in 1 2 3; do
  python tools/siren_step_time.py 10 2>/dev/null | grep median
  ISO_SIREN_MERGED=0 python tools/siren_step_time.py 10 2>/dev/null | grep median | sed 's/default /two-launch/'
done
