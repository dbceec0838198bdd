"""Writes tests/golden/graph_config.json.

Source of the expected values: SURVEY.md section 8(a) row L -- outputs of the reference's own
src/ggnn/base/graph_config.cpp recorded by the surveyor in the authoring container (the
reference TU cannot be built in this image without stand-ins for glog, so it is not rebuilt
here).  The values below are transcribed from that row; this script only formats them.
"""
import json
import os

GOLDEN = [
    # N, D, KBuild -> fields recorded in SURVEY.md 8(a) row L
    dict(N=10_000, D=128, KBuild=24, G=7, S0=29, S0_off=53, SG=4, SG_off=4, N_all=11_824,
         ST_all=1_824),
    dict(N=1_000_000, D=128, KBuild=24, G=32, S0=30, S0_off=16_960, SG=1, SG_off=0,
         N_all=1_033_824, ST_all=33_824, Ns=[1_000_000, 32_768, 1_024, 32]),
    dict(N=1_000_000, D=960, KBuild=24, G=32, S0=30, S0_off=16_960, SG=1, SG_off=0,
         N_all=1_033_824, ST_all=33_824, Ns=[1_000_000, 32_768, 1_024, 32]),
    dict(N=12_500_000, D=96, KBuild=24, G=73, S0=32, S0_off=51_456, SG=0, SG_off=32,
         N_all=12_672_896),
    dict(N=125_000_000, D=128, KBuild=24, G=157, S0=32, S0_off=1_163_424, SG=0, SG_off=32,
         N_all=125_793_824),
    dict(N=25_000, D=128, KBuild=24, G=9, S0=34, S0_off=214, SG=3, SG_off=5),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_config.json")
    with open(out, "w") as f:
        json.dump(GOLDEN, f, indent=1)
    print("wrote", out)
