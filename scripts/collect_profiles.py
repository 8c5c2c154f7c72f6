#!/usr/bin/env python3
"""Copies the judged evidence of a profile_round.sh run (gpurun_out/<tag>_*) into profiles/ under round-stable names:
python scripts/collect_profiles.py r02f r02"""
import json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
g, p = "gpurun_out", "profiles"
def cp(src, dst):
  if os.path.exists(os.path.join(g, src)):
    shutil.copy(os.path.join(g, src), os.path.join(p, dst)); print(dst)
  else:
    print("missing", src)
cp("%s_bench.json" % tag, "%s_bench_bs32_256_bf16.json" % rnd)
cp("%s_prof/%s_kernel_stats.csv" % (tag, tag), "%s_bench_bs32_256_bf16_kernel_stats.csv" % rnd)
cp("%s_prof1/%s_single_kernel_stats.csv" % (tag, tag), "%s_bench_bs32_256_bf16_single_stream_kernel_stats.csv" % rnd)
cp("%s_pmc_hbm_traffic.json" % tag, "%s_pmc_hbm_traffic.json" % rnd)
cp("%s_pmc_fetch_per_kernel.csv" % tag, "%s_pmc_fetch_per_kernel.csv" % rnd)
cp("%s_pmc_write_per_kernel.csv" % tag, "%s_pmc_write_per_kernel.csv" % rnd)
