for cfg in "--dim 1280 --rows 24 --nl 35 --npsflin 3 --steps 10" "--dim 512 --rows 50 --nl 35 --npsflin 3 --steps 20" "--dim 1024 --rows 24 --nl 35 --npsflin 2 --steps 10"; do
  for m in 0 1; do
    echo "== $cfg ndir-mfma=$m"
    MPSFR_OTF_MFMA_NDIR=$m python scripts/variants.py run --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 $cfg 2>&1 | grep default
  done
done
