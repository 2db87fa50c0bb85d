for g in 0 1; do echo "== yield graph=$g"; SCDA_GAN_GRAPH=$g python scripts/host_cpu_use.py yield 2>/dev/null | tail -6; done
