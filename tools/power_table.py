"""Energy per image by consumer: counts from the rocprofv3 --pmc passes (profiles/rN_pmc.json, fp16 slots,
one launch of `images` images per kernel) x the energy per event measured with tools/power_probe.sh on the same kind of box
(single instruction / access streams: package power above the clocked-but-quiet floor, divided by the event rate), against the package power and time
of the forward itself (tools/clock_probe.sh).  Everything that is not attributed is the time-proportional remainder.

usage: python tools/power_table.py profiles/r4_pmc.json <images per launch> <forward W> <forward images/s> <out.json>"""
import json
import sys

pmc, images, watts, ips, out = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), sys.argv[5]
# The floor: what the package draws with every CU clocked and executing, but next to nothing switching -- the v_exp_f32 stream
# of the probe (quarter-rate, no operands from memory): 529 W at 2.4 GHz, of which ~30 W are the instructions themselves.  (The
# card IDLES at 250 W with the shader clock at 0.3 GHz: an energy per event taken above THAT would count the clock tree and
# the uncore once per stream -- the round-3 table did, and its consumers summed to more than the package.)
FLOOR_W = 500.0
# nJ per wave-instruction / per LDS-array cycle, pJ per byte: (package W of the single stream - FLOOR_W) / event rate,
# profiles/r4_power_probe.txt (one box, one call, the forward measured in the same call)
E = {"mfma_16x16x32": 7.12,    # 1345 W at 2.01 GHz, 1.187e11 /s
     "mfma_32x32x16": 15.6,    # 1306 W at 1.79 GHz (!), 5.164e10 /s
     "valu": 0.646,            # v_fma_f32: 1133 W at 2.40 GHz, 9.793e11 /s
     "lds_cycle": 1.087,       # ds_read_b128 saturating the LDS array (256 B/clk/CU): 1147 W, 1.488e11 reads/s x 4 array cycles
     "l2_byte_pj": 10.5,       # global_load_dwordx4 from an L2-resident 1 MB window (L1 thrashed): 647 W at 14.0 TB/s
     "hbm_byte_pj": 100.6}     # the same from a 4 GB buffer: 976 W at 4.73 TB/s (L2 + fabric + HBM3E)
slots = json.load(open(pmc))["slots"]["fp16"]
rows, tot = {}, {k: 0.0 for k in ("mfma", "valu", "lds", "l2", "hbm")}
for name, s in sorted(slots.items()):
    if not name.startswith("stage") or "valu_insts" not in s:
        continue
    big = name.startswith(("stage1_", "stage2_")) and not name.endswith("_se")      # v_mfma_f32_32x32x16_f16 kernels
    hbm_b = s.get("hbm_bytes_per_launch", 0.0)
    l2_b = max(0.0, s.get("l1_to_l2_read_requests", 0.0) * 64.0 - 0.0)                # TCP_TCC_READ_REQ, 64 B per request
    r = {"mfma": s["mfma_insts"] * (E["mfma_32x32x16"] if big else E["mfma_16x16x32"]) * 1e-9,
         "valu": s["valu_insts"] * E["valu"] * 1e-9,
         "lds": s.get("lds_idx_active_cycles", 0.0) * E["lds_cycle"] * 1e-9,
         "l2": l2_b * E["l2_byte_pj"] * 1e-12,
         "hbm": hbm_b * E["hbm_byte_pj"] * 1e-12}
    rows[name] = {k: v / images for k, v in r.items()}
    rows[name]["l1_to_l2_read_bytes_per_image"] = l2_b / images
    rows[name]["hbm_bytes_per_image"] = hbm_b / images
    for k in tot:
        tot[k] += r[k] / images
j_image = watts / ips
res = {"source": {"pmc": pmc, "images_per_launch": images, "forward_package_w": watts, "forward_images_per_s": ips,
                  "floor_w": FLOOR_W, "energy_per_event": E},
       "joule_per_image": {"total": j_image, **tot, "clocked_floor": FLOOR_W / ips,
                           "unattributed_time_proportional": j_image - sum(tot.values()) - FLOOR_W / ips},
       "share": {k: v / j_image for k, v in tot.items()},
       "watts_at_this_speed": {k: v * ips for k, v in tot.items()},
       "per_kernel_joule_per_image": rows}
res["share"]["clocked_floor"] = FLOOR_W / ips / j_image
res["share"]["unattributed_time_proportional"] = res["joule_per_image"]["unattributed_time_proportional"] / j_image
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("joule_per_image", "share", "watts_at_this_speed")}, indent=1))
