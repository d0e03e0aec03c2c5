#!/usr/bin/env python3
"""Turns the text the stand-alone microbenchmarks print (hop_probe, hop_bench, chain_probe, ingest_bench, mfma16_clock,
mfma_clock, one after the other, each behind a line `== name`) into the JSON files bench.py reads its hand-off cost
and matrix-core clock from.  Usage: micro_summary.py <text file> <tag>  ->  profiles/<tag>_hop.json, <tag>_ingest.json,
<tag>_mfma_clock.json"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
sections, name = {}, None
for line in open(src):
  line = line.rstrip("\n")
  if line.startswith("== "):
    name = line[3:].strip()
    sections[name] = []
  elif name:
    sections[name].append(line)

def table(lines, pattern):
  out = {}
  for ln in lines:
    m = re.match(pattern, ln)
    if m:
      out[m.group(1).strip()] = float(m.group(2))
  return out

rel = os.path.relpath(src, ROOT)
gathers = table(sections.get("hop_probe", []), r"all-gather (.+?)\s+([0-9.]+) us/round")
pingpong = table(sections.get("hop_probe", []), r"ping-pong (.+?)\s+one way\s+([0-9.]+) ns")
loads = table(sections.get("hop_probe", []), r"dependent agent-scope load, (.+?)\s+([0-9.]+) ns")
chains = table(sections.get("chain_probe", []), r"(.+?)\s+([0-9.]+) us/cycle")
rounds = table(sections.get("hop_bench", []), r"(.+?)\s+([0-9.]+) us/round")
group_to_all = gathers.get("32x8 R=8 agent, uncached")
hop = {
  "source": rel, "programs": ["scripts/hop_probe.hip", "scripts/hop_bench.hip", "scripts/chain_probe.hip"],
  "dependent_agent_scope_load_ns": loads, "one_producer_one_consumer_one_way_ns": pingpong,
  "all_gather_us_per_round": gathers, "all_gather_hop_bench_us_per_round": rounds, "chains_us_per_cycle": chains,
  # the cost of one hand-off as the batch-1 decoder makes them: a short vector published by a small group of workgroups
  # (32 x 8 words, 8 replicas) and consumed by all 256, store to consumed, rounds chained
  "hop_us": group_to_all, "hop_us_is": "all-gather 32x8 R=8 agent, uncached (hop_probe): us per dependent round",
  "five_hop_chain_without_work_us": chains.get("5 hops: same with ah consumed by 8 only (+ final all)"),
}
json.dump(hop, open(os.path.join(ROOT, "profiles", f"{tag}_hop.json"), "w"), indent=1)
ingest = {"source": rel, "program": "scripts/ingest_bench.hip",
          "us_per_round": table(sections.get("ingest_bench", []), r"(.+?)\s+([0-9.]+) us per round"),
          "note": "all 256 CUs take in one 64 KB vector; rows marked GAVE UP in the text are runs whose bounded waits expired"}
json.dump(ingest, open(os.path.join(ROOT, "profiles", f"{tag}_ingest.json"), "w"), indent=1)
m16, m32 = [], []
for ln in sections.get("mfma16_clock", []):
  m = re.match(r"(\d+) chain\(s\), blocks\s+(\d+):\s+([0-9.]+) ns per MFMA, shader clock ([0-9.]+) GHz, ([0-9.]+) cycles", ln)
  if m:
    m16.append(dict(chains=int(m.group(1)), blocks=int(m.group(2)), ns_per_mfma=float(m.group(3)), ghz=float(m.group(4)), cycles=float(m.group(5))))
for ln in sections.get("mfma_clock", []):
  m = re.match(r"blocks\s+(\d+),\s+(\d+) MFMAs/wave: .* =\s+([0-9.]+) ns per MFMA, shader clock ([0-9.]+) GHz, ([0-9.]+) cycles", ln)
  if m:
    m32.append(dict(blocks=int(m.group(1)), mfmas_per_wave=int(m.group(2)), ns_per_mfma=float(m.group(3)), ghz=float(m.group(4)), cycles=float(m.group(5))))
full = [r for r in m16 if r["blocks"] == 256]
clock = {"source": rel, "programs": ["scripts/mfma16_clock.hip", "scripts/mfma_clock.hip"],
         "v_mfma_f32_16x16x4_f32": m16, "v_mfma_f32_32x32x2_f32": m32,
         "sustained_ghz_all_cus": min(r["ghz"] for r in full) if full else None,
         "cycles_per_16x16x4_f32": full[0]["cycles"] if full else None,
         # 256 CUs x 4 SIMDs x 2048 flop per 16x16x4 product / (cycles / clock)
         "sustained_f32_mfma_tflops": round(1024 * 2048 / (full[0]["ns_per_mfma"] * 1e-9) / 1e12, 1) if full else None}
json.dump(clock, open(os.path.join(ROOT, "profiles", f"{tag}_mfma_clock.json"), "w"), indent=1)
print(json.dumps({"hop_us": hop["hop_us"], "five_hop_chain": hop["five_hop_chain_without_work_us"],
                  "sustained_f32_mfma_tflops": clock["sustained_f32_mfma_tflops"]}))
