# exact mode per-table kernel with shifted LDS copies for the odd offsets (timing experiment)
for sh in 0 1 0 1; do
  echo "== VND_SPEC_SHIFT=$sh"
  VND_SPEC_SHIFT=$sh timeout -k 10 200 python tools/exact_try.py 2>&1 | grep -E "^(generic ordered|spec exact  |spec exact nt=256 rr=4|spec exact nt=128 rr=4)" | head -4 | cut -c1-150
  VND_SPEC_SHIFT=$sh timeout -k 10 200 python tools/exact_class_try.py 2>&1 | grep -E "^(spec exact  |spec exact nt=256)" | head -2 | cut -c1-150
done
