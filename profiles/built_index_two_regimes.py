import sys, json
sys.path.insert(0, 'profiles'); sys.path.insert(0, '.')
import built_index_probe as b
for nt in (256, 4096):
    r = b.run(1_000_000, 128, 2, nt, policies=((2, 0.45, 1024, 100),), phases=False, parity_queries=8)
    sr = r["search_thr0.45"]
    print(nt, sr["queries_per_sec"], sr["ms_per_step"], sr["recall_at_5"], sr["surviving_centroids"], sr["candidates"], sr["stage_ms"], sr.get("parity"), r["build_index_s"])
