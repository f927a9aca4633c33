timeout -k 10 200 python tools/exact_try.py 2>&1 | grep -E "^(generic ordered|spec exact)" | head -7 | cut -c1-150
timeout -k 10 300 python tools/fanout_spec_try.py 2>&1 | grep exact | tail -4 | cut -c1-150
