for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-parity --no-train --no-frontend --cpu-tokens 0 2>/dev/null | cut -c1-160; done
