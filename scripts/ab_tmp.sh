for b in 4096 512 4096 512 384 768; do echo "== SCDA_ADAM_BLOCKS=$b"; SCDA_ADAM_BLOCKS=$b python scripts/device_phase_times.py 2>/dev/null | tail -5; done
for b in 320 384 448 512 640 768; do SCDA_ADAM_BLOCKS=$b python scripts/time_adam.py 2>/dev/null | tail -1; done
