#!/bin/bash
# round 3, GPU call 31: the clip's noise draws — host generator (fp16 / fp32 on the host cores + copy) vs device generator — and
# bench.py with the device generator (the reference CLI's choice) twice at a tiny shape: same digest
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r3_noise_draw_host_vs_device_generator.log
timeout 40 python - > $L 2>&1 <<'PY'
import time, torch
dev = torch.device("cuda:0")
shapes = [(1, 3, 8, 320, 320), (1, 4, 8, 320, 320)]
torch.randn(8, device=dev); torch.cuda.synchronize()
for name, mk, dt in (("host generator fp16 (rounds 1-3 bench)", lambda: torch.Generator().manual_seed(1), torch.float16),
                     ("host generator fp32", lambda: torch.Generator().manual_seed(1), torch.float32),
                     ("device generator fp16 (reference CLI)", lambda: torch.Generator(device=dev).manual_seed(1), torch.float16)):
    ts = []
    for rep in range(3):
        g = mk(); torch.cuda.synchronize(); t = time.perf_counter()
        xs = [torch.randn(s, generator=g, device=g.device, dtype=dt).to(dev) for s in shapes]
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{name}: {min(ts):.2f} ms per clip (best of 3), host threads {torch.get_num_threads()}")
PY
cat $L
for i in 1 2; do
  timeout 60 python bench.py --steps 1 --warmup 0 --height 64 --width 64 --ddim-steps 2 --no-cpu-baseline --no-kernel-events --text-encoder standin --digest 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench 64x64 device generator run $i sha', d['config']['output_sha256'][:16])" | tee -a $L
done
