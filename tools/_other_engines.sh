for e in bf16x6 f32; do echo "== SS_CONV_ENGINE=$e"; SS_CONV_ENGINE=$e timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3; done
