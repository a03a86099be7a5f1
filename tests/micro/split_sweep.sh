# diagnostic: GPU test suite + bench line for every split factor of the fused kernel
for m in ${SPLITS:-1 2 4 8}; do
  echo "=== GATRES_FUSED_SPLIT=$m"
  GATRES_FUSED_SPLIT=$m timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -4
  GATRES_FUSED_SPLIT=$m timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done
