#!/usr/bin/env python3
"""decode_layer_table.py — the per-layer table of a decode step from a rocprofv3 kernel trace, per launch SHAPE rather than per template instantiation.

    python tools/decode_layer_table.py <dir with *_kernel_trace.csv> [out.md]

rocprofv3's --stats table merges launches that share a kernel instantiation: k_mmvq_stream<12, 12, true, false> is wo, norm + QKV (Q4_K wv) and the Q4_K down
projection at once. A decode token is a fixed sequence of launches, so a launch's ROLE follows from its neighbours in the trace:
    a streamed mat-vec followed by k_attn_decode                      -> norm + QKV (+ RoPE, KV store)
    ... preceded by k_attn_decode                                     -> wo + residual
    the GLU instantiation (<.., true, true>)                          -> norm + gate/up + SwiGLU
    ... preceded by the GLU launch                                    -> down + residual
    ... neither before attention nor after attention / GLU            -> norm + lm_head (the last mat-vec of the token)
For every role: calls, average duration (the dispatch's begin -> end, what --stats averages) and the average idle time to the NEXT dispatch on the queue; then
the per-layer sum, quoted verbatim in DESIGN.md. Only dispatches of decode steps are used: the trace is cut to the longest run in which every k_attn_decode is
followed, five mat-vec launches later, by the next one (prompt passes and the profiling passes of bench.py have other sequences)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True) + glob.glob(d + "/*kernel_trace.csv")
rows = []
for f in sorted(set(files)):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("mi355x::", "").replace("void ", "")))
rows.sort()
is_mv = lambda n: n.startswith("k_mmvq_stream<") or n.startswith("k_mmvq_fused<")
is_glu = lambda n: is_mv(n) and n.split(">")[0].rstrip().endswith("true")
is_attn = lambda n: n.startswith("k_attn_decode<")

roles = [None]*len(rows)
# neighbours among the "layer" kernels only (mat-vecs and attention); small element kernels between them are ignored
layer_idx = [i for i, r in enumerate(rows) if is_mv(r[2]) or is_attn(r[2])]
for q, i in enumerate(layer_idx):
    n = rows[i][2]
    prev = rows[layer_idx[q - 1]][2] if q > 0 else ""
    nxt = rows[layer_idx[q + 1]][2] if q + 1 < len(layer_idx) else ""
    if is_attn(n): roles[i] = "attention"
    elif is_glu(n): roles[i] = "norm + gate/up + SwiGLU"
    elif is_attn(nxt): roles[i] = "norm + QKV (+RoPE, KV store)"
    elif is_attn(prev): roles[i] = "wo + residual"
    elif is_glu(prev): roles[i] = "down + residual"
    else: roles[i] = "norm + lm_head"
# keep decode tokens only: a token = ... [QKV attn wo GLU down] x n_layer, lm_head; drop everything outside complete (QKV, attention, wo, GLU, down) quintuples
keep = [False]*len(rows)
want = ["norm + QKV (+RoPE, KV store)", "attention", "wo + residual", "norm + gate/up + SwiGLU", "down + residual"]
q = 0
while q + 4 < len(layer_idx):
    if [roles[layer_idx[q + j]] for j in range(5)] == want:
        for j in range(5): keep[layer_idx[q + j]] = True
        if q + 5 < len(layer_idx) and roles[layer_idx[q + 5]] == "norm + lm_head": keep[layer_idx[q + 5]] = True
        q += 5
    else:
        q += 1
agg = defaultdict(lambda: [0, 0.0, 0.0, defaultdict(int)])
for i, r in enumerate(rows):
    if not keep[i]: continue
    a = agg[roles[i]]
    a[0] += 1; a[1] += (r[1] - r[0])/1e3
    if i + 1 < len(rows): a[2] += max(0.0, (rows[i + 1][0] - r[1])/1e3)
    a[3][r[2].split("(")[0]] += 1
n_layers = agg[want[1]][0]
n_tok = max(1, agg["norm + lm_head"][0])
out = []
out.append(f"decode steps in the trace: {n_tok}; layers per step: {n_layers/n_tok:.0f}")
out.append("")
out.append("| launch (role) | instantiations | calls | avg duration us | avg idle to the next dispatch us |")
out.append("|---|---|---:|---:|---:|")
tot = 0.0; tot_gap = 0.0
for role in want + ["norm + lm_head"]:
    a = agg[role]
    if not a[0]: continue
    inst = ", ".join(f"{k} x{v}" for k, v in sorted(a[3].items(), key=lambda kv: -kv[1]))
    out.append(f"| {role} | {inst} | {a[0]} | {a[1]/a[0]:.2f} | {a[2]/a[0]:.2f} |")
    if role != "norm + lm_head": tot += a[1]/a[0]; tot_gap += a[2]/a[0]
out.append("")
out.append(f"per-layer sum of the five launches' average durations: {tot:.2f} us (+ {tot_gap:.2f} us idle between dispatches = {tot + tot_gap:.2f} us per layer)")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
