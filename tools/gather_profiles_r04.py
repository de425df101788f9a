#!/usr/bin/env python3
"""Copy round 4's measurement files from gpurun_out/ (scratch, merged back from the MI355X box by gpurun) into profiles/ under judged names.
The rocprofv3 summaries come from tools/summarise_profiles_r03.py <tag>; this script takes the rest: hand-off measurements, A/B tables, CPU
profiles, the GPU test tier and the bench lines.   usage: gather_profiles_r04.py r04_v1"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_v1"


def cp(src, name):
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copy(s, os.path.join(P, "%s_%s" % (tag, name)))
    else:
        print("missing", src)


def cat(parts, name, head=None):
    out = []
    if head:
        out.append(head.rstrip() + "\n")
    for title, src in parts:
        s = os.path.join(G, src)
        if not os.path.exists(s):
            print("missing", src)
            continue
        out.append("\n# ---- %s  (gpurun_out/%s)\n" % (title, src))
        out.append(open(s, errors="replace").read().rstrip() + "\n")
    open(os.path.join(P, "%s_%s" % (tag, name)), "w").write("".join(out))


# the hand-off, first measurements (measured first) and the product's round trip as it developed
cat([("launch + synchronise / launch + flag / resident mailbox, copy-only device work (tools/micro/handoff)", "r04_a/handoff.txt"),
     ("the prize: cycle counters in the seams of the bound encoder, jobs off (X265HIP_DEBUG_CUTIME=1)", "r04_a/cutime_1080p_medium.txt"),
     ("host of the GPU box", "r04_a/host.txt")], "handoff_and_prize.txt",
    "# round 4, first GPU call: what a synchronous hand-off costs and what the residual quad-trees cost the host")
cat([("resident server, first version (mailbox in host memory)", "r04_e/cuserve_rt_mode0_3.txt"),
     ("final tree, mailbox half in device memory (default)", "r04_z/cuserve_rt_mailbox_device.txt"),
     ("final tree, X265HIP_CUSERVE_MAILBOX=host", "r04_z/cuserve_rt_mailbox_host.txt"),
     ("final tree, one launch per job (X265HIP_CUSERVE_MODE=1)", "r04_z/cuserve_rt_mode1.txt")], "cuserve_rt.txt",
    "# tools/micro/cuserve_rt: the product's CU job round trip through the C ABI (submit -> first luma unit's forward half -> every unit's inverse half)")
cp("r04_z/cuserve_rt_stamps.txt", "cuserve_rt_stamps.txt")
cat([("host memory (kind 0) vs device memory written through the BAR (1 hipMalloc, 2 fine-grained, 3 uncached), 6 KB payload", "r04_l/bar_mailbox_0.txt"),
     ("", "r04_l/bar_mailbox_1.txt"), ("", "r04_l/bar_mailbox_2.txt"), ("", "r04_l/bar_mailbox_3.txt"),
     ("breakdown: device part vs transport, 64 B and 3 KB payloads, with and without an HDP flush by the host", "r04_n/bar_mailbox_breakdown.txt")], "bar_mailbox_breakdown.txt",
    "# tools/micro/bar_mailbox: can the host write a mailbox in device memory, and what does the hand-off gain")
cat([("hipFree and friends beside a resident kernel; which host memory a resident kernel sees", "r04_c/mailbox_diag_memory.txt"),
     ("first diagnosis run", "r04_b/mailbox_diag.txt")], "mailbox_diag.txt", "# tools/micro/mailbox_diag")
cat([("before the stream pool", "r04_l/create_cost.txt"), ("final tree", "r04_z/create_cost.txt")], "create_cost.txt", "# tools/micro/create_cost: per-picture device objects")
cp("r04_j/pin_thp.txt", "pin_thp.txt")
cat([("before (every source picture created its own stream: 4-10 ms each)", "r04_j/startup_marks.txt"), ("final tree", "r04_z/startup_marks.txt")], "startup_marks.txt",
    "# X265HIP_DEBUG_STARTUP=1 marks of the bound encoder, 1080p medium 120 frames")
# A/B tables, in the order the features arrived
cat([("resident server vs launches, first binding (transforms only)", "r04_b/ab.txt"), ("after the first fixes", "r04_c/ab.txt"), ("", "r04_d/ab.txt"),
     ("start / leave races fixed, no L2-invalidating fences; row batching", "r04_e/ab.txt"), ("stage 2: distortions and psy out of the jobs", "r04_f/ab.txt"), ("", "r04_g/ab.txt"),
     ("sub-pel SATD tables and rectangular PUs", "r04_i/ab1080.txt"), ("", "r04_j/ab1080.txt"),
     ("coded psy energy on the device; page-locked buffers on huge pages (hippin = hipHostMalloc)", "r04_k/ab1080.txt"),
     ("the CU's final sse / psy from the units (dist1 = without)", "r04_l/ab1080.txt"),
     ("mailbox half in device memory (hostbox = in host memory); dead sub_ps / add_ps put off (dist2 = without)", "r04_m/ab1080.txt"),
     ("16x16 CUs as jobs again (min16)", "r04_p/ab1080.txt"), ("jobs submitted ahead at the skip evaluation (spec0 = without)", "r04_q/ab1080.txt"),
     ("SAD-surface rows waiting for company: 2 / 8 / 16 bands", "r04_s/ab1080.txt"), ("resident server workgroups: 32 (default on this box) / 64 / 20", "r04_t/ab1080.txt"),
     ("FINAL TREE: on vs the round-3 feature set (r3: jobs, sub-pel tables, rectangular PUs, row batching, huge pages off)", "r04_z/ab1080.txt")], "ab_1080p_medium.txt",
    "# tools/ab_encode.py, 1080p medium 120 frames, interleaved rounds, byte-identity against the unmodified reference checked in every run (column `identical`)")
cp("r04_c23/configs2_4k_slow_star_ab.txt", "configs2_4k_slow_star_ab.txt")
cp("r04_c23/configs3_4k_main10_slower_ab.txt", "configs3_4k_main10_slower_ab.txt")
cp("r04_f/subpel_hit.txt", "subpel_hit.txt")
cat([("ioctl / munmap / mmap", "r04_f/ioctl_callers.txt"), ("runtime libraries", "r04_f/runtime_callers.txt")], "binding_overhead_callers.txt",
    "# CPU sampler with call chains (tools/prof/cpusample.c X265HIP_CPUSAMPLE_STACK=1, tools/prof/callers.py): who calls the runtime's system calls")
cp("r04_z/cpu_profile_bound_encoder.txt", "cpu_profile_bound_encoder.txt")
cp("r04_e/cpu_profile_hip.txt", "cpu_profile_bound_encoder_first_jobs.txt")
cp("r04_z/gpu_test_tier.txt", "gpu_test_tier.txt")
cp("r04_z/smoke.txt", "smoke.txt")
cp("r04_z/bench_line.json", "bench_line.json")
cp("r04_z/bench_line_driver_args.json", "bench_line_driver_args.json")
# the served-by-GPU lines of the final A/B (waits by site, jobs ahead, put-off calls, SAD surfaces)
try:
    d = json.load(open(os.path.join(G, "r04_z", "ab1080.json")))
    with open(os.path.join(P, "%s_encode_served_lines.txt" % tag), "w") as f:
        for k, v in d["configs"].items():
            f.write("# ---- configuration %s: %s\n" % (k, v.get("env")))
            for l in v["served"]:
                f.write(l + "\n")
except OSError:
    print("missing r04_z/ab1080.json")
print(sorted(x for x in os.listdir(P) if x.startswith(tag)))
