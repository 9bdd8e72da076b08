"""One line per config of a bench.py `configs12` object: python scripts/show_configs12.py file.json [tag]"""
import json, sys
o = json.load(open(sys.argv[1]))
o = o.get("configs12", o)
tag = sys.argv[2] if len(sys.argv) > 2 else ""
for k in ("config1_type0", "config2_mixed"):
    c = o[k]
    print(f"{tag:14s} {k:14s} resident {c['resident_ms']:.3f} ms (min {c['resident_ms_min']:.3f}, device {c['device_ms']:.3f})  host->host {c['host_to_host_ms']:.3f} ms "
          f"(min {c['host_to_host_ms_min']:.3f})  step frac {c['step_frac_of_mfma_peak']:.3f}  filter frac {c['filter_kernel_frac']:.3f}  retried/call {c['retry_queries_per_call']:.2f}")
print(f"{tag:14s} load_data {o['load_data_ms']:.1f} ms (second call {o['load_data_ms_second_call']:.1f})")
