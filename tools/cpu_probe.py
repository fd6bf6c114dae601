import os, sys, time, json, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
d3 = {"text": {"vocab_size": 151936, "hidden_size": 2048, "intermediate_size": 11008, "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2, "rms_norm_eps": 1e-6, "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
      "vision": {"depth": 32, "hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "in_channels": 3, "patch_size": 14, "spatial_merge_size": 2, "temporal_patch_size": 2, "window_size": 112, "out_hidden_size": 2048, "fullatt_block_indexes": [7, 15, 23, 31]}, "tie_word_embeddings": True}
for th in (16, 32, 64, 128):
    os.environ["IADR1_CPU_THREADS"] = str(th)
    t0 = time.time()
    r = bench.cpu_baseline(d3, 8.0)
    print(th, round(time.time() - t0, 1), r["sample"][-80:], flush=True)
