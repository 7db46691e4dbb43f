#!/usr/bin/env python3
"""Dev helper: compile one translation unit to gfx950 assembly and print, per kernel matching a pattern, the
instruction mix of every loop block (VALU / fp64 / LDS / VMEM / waits / moves) plus register usage."""
import re, subprocess, sys, os
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "tick_wg_kernel"
out = "/tmp/asm/" + os.path.basename(src) + ".s"
os.makedirs("/tmp/asm", exist_ok=True)
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                    "-Rpass-analysis=kernel-resource-usage", "-o", out, src], capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr[-3000:])
rp = r.stderr
s = open(out).read()
for m in re.finditer(r'^(_Z\w+):.*?\.end_amdhsa_kernel', s, re.S | re.M):
    name = m.group(1)
    if pat not in name: continue
    body = m.group(0)
    res = re.search(re.escape(name) + r'.*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)', rp, re.S)
    sg = re.search(re.escape(name) + r'.*?TotalSGPRs: (\d+)', rp, re.S)
    print("==", name[:110], "VGPR", res and res.group(1), "scratch", res and res.group(2), "SGPR", sg and sg.group(1))
    cur = None
    for l in body.split('\n'):
        t = l.strip()
        mm = re.match(r'^(\.LBB\d+_\d+):(.*)', t)
        if mm:
            if cur and cur['n'] >= 30 and 'Loop' in cur['note']: print("  ", cur)
            cur = {'name': mm.group(1), 'note': mm.group(2).strip()[2:44], 'n': 0, 'valu': 0, 'f64': 0, 'ds': 0, 'vmem': 0, 'salu': 0, 'wait': 0, 'mov': 0}
        elif cur is not None and t and not t.startswith(';') and not t.startswith('.'):
            op = t.split()[0]; cur['n'] += 1
            if op.startswith('v_'): cur['valu'] += 1
            if 'f64' in op: cur['f64'] += 1
            if op.startswith('ds_'): cur['ds'] += 1
            if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur['vmem'] += 1
            if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_cbranch', 's_branch', 's_nop')): cur['salu'] += 1
            if op.startswith('s_waitcnt'): cur['wait'] += 1
            if op.startswith('v_mov'): cur['mov'] += 1
    if cur and cur['n'] >= 30 and 'Loop' in cur['note']: print("  ", cur)
