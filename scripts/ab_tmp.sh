echo "== glds"; python scripts/bench_conv_layers.py 2>/dev/null | cut -c1-12,68-110
echo "== SCDA_WGRAD_NO_GLDS=1"; SCDA_WGRAD_NO_GLDS=1 python scripts/bench_conv_layers.py 2>/dev/null | cut -c1-12,68-110
