echo "--- partial-frame stores through L2 (new default)"; timeout -k 10 300 python tools/c8_try.py 2>&1 | grep -E "^(generic|spec default|spec nt=128)" | head -3
echo "--- non-temporal (old)"; VND_FORCE_NT=1 timeout -k 10 300 python tools/c8_try.py 2>&1 | grep -E "^(generic|spec default|spec nt=128)" | head -3
