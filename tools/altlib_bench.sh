#!/bin/bash
# bench.py (no extras) + large-batch kernel time with each alternative build in altlib/*.so (tuning experiments; GPU box)
echo "main: $(python bench.py --no-extras | cut -c1-120 | sed 's/.*"value": \([0-9]*\).*"ms_per_step": \([0-9.]*\).*/\1 \2/') | $(python tools/quick_time.py 1048576 2>&1 | tail -1)"
for L in altlib/*.so; do
  echo "$(basename $L): $(tools/with_altlib.sh $L python bench.py --no-extras | cut -c1-120 | sed 's/.*"value": \([0-9]*\).*"ms_per_step": \([0-9.]*\).*/\1 \2/') | $(tools/with_altlib.sh $L python tools/quick_time.py 1048576 2>&1 | tail -1)"
done
