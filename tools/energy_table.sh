#!/bin/bash
# tools/energy_table.sh <round> -- the headline priced in joules (round-5 review, item 5): every variant is a bench.py run of >= 2 s of the same
# steps; bench.py's own rocm-smi sampler (2.5 s of the same steps behind the timed region: `roofline.hwmon`) gives the socket power and the
# shader clock, the timed region the ms per step; mJ per step = W x ms.  Variants: the XCD super-tile of the GEMMs (group=TxN), the
# best-density matrix of the GMM leg (u32 all states | u8 | of the aligned state only), streaming stores plain instead of non-temporal
# (a second build of the library, -DAMX_PLAIN_STORES), both contracts, the bf16 arithmetic.  L2 hit rates of the group variants from a
# separate --pmc pass (kernel-trace only).  Writes gpurun_out/<round>/energy/*.json + energy.json.
round=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$round/energy
mkdir -p $out
cd $root
run() {  # name, env-assignments (or -), bench args...
    name=$1; envs=$2; shift 2
    if [ "$envs" = "-" ]; then envs=""; fi
    env $envs python bench.py "$@" --no-cpu-baseline --no-configs 2>$out/$name.err | grep '^{"metric"' | tail -1 > $out/$name.json
    [ -s $out/$name.json ] || echo "energy_table: $name produced no line: $(tail -1 $out/$name.err)"
}
PL=tools/build/librasr_amd_plain.so
run pipeline_default - --steps 60 --warmup 5
run pipeline_group_8x8 - --steps 60 --warmup 5 --nn-tuning group=8x8
run pipeline_group_2x16 - --steps 60 --warmup 5 --nn-tuning group=2x16
run pipeline_group_32x4 - --steps 60 --warmup 5 --nn-tuning group=32x4
run pipeline_best_u8 - --steps 60 --warmup 5 --best-density u8
run pipeline_best_aligned - --steps 60 --warmup 5 --best-density aligned
run pipeline_contract_off - --steps 60 --warmup 5 --contract off
run pipeline_bf16 - --steps 60 --warmup 5 --precision bf16
[ -f $PL ] && run pipeline_plain_stores AMX_LIBRARY=$PL --steps 60 --warmup 5
run nn_default - --workload nn-pipeline --steps 100 --warmup 5
run nn_group_8x8 - --workload nn-pipeline --steps 100 --warmup 5 --nn-tuning group=8x8
run nn_group_2x16 - --workload nn-pipeline --steps 100 --warmup 5 --nn-tuning group=2x16
run nn_group_32x4 - --workload nn-pipeline --steps 100 --warmup 5 --nn-tuning group=32x4
[ -f $PL ] && run nn_plain_stores AMX_LIBRARY=$PL --workload nn-pipeline --steps 100 --warmup 5
# L2 hit rate of the GEMM launches per group shape (counters in their own pass)
cd /tmp && export TMPDIR=/tmp
for g in default 8x8 2x16 32x4; do
    rm -rf /tmp/en_pmc_$g
    tun=""; [ "$g" != "default" ] && tun="--nn-tuning group=$g"
    AMX_BENCH_NO_SMI=1 timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/en_pmc_$g -- python $root/bench.py --workload nn-pipeline --steps 3 --warmup 1 --no-cpu-baseline --no-configs $tun > /tmp/en_pmc_$g.log 2>&1
    f=$(find /tmp/en_pmc_$g -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $root/tools/pmc_summary.py $f gemm_mx > $out/l2_group_$g.txt
done
cd $root
python - "$out" <<'PY'
import glob, json, os, re, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(f)[:-5]
    if name == "energy":
        continue
    try:
        d = json.load(open(f))
    except Exception:
        continue
    r = d.get("roofline") or {}
    hw = r.get("hwmon") or {}
    sec = r.get("second") or {}
    def launch_ms(x):
        return round(x.get("avg_launch_ms", 0) * x.get("launches", 0) / max(1, min(d["steps"], 20)), 4) if x else None
    row = dict(frames_per_s=d["value"], ms_per_step=d["ms_per_step"], socket_power_W=hw.get("socket_power_W"), sclk_GHz=hw.get("sclk_GHz"),
               joules_per_step=round(hw["socket_power_W"] * d["ms_per_step"] * 1e-3, 4) if hw.get("socket_power_W") else None,
               dominant_kernel=(r.get("kernel") or "")[:40], dominant_ms_per_step=r.get("summed_ms_per_step") or launch_ms(r),
               second_kernel=(sec.get("kernel") or "")[:40], second_ms_per_step=sec.get("summed_ms_per_step") or launch_ms(sec),
               build=d.get("build", "")[-20:])
    g = name.split("group_")[-1] if "group_" in name else ("default" if name in ("nn_default",) else None)
    if name.startswith("nn_") and g:
        p = os.path.join(out, "l2_group_%s.txt" % g)
        if os.path.exists(p):
            hit = miss = 0.0
            for line in open(p):
                m = re.match(r"\s+(TCC_HIT_sum|TCC_MISS_sum)\s+([0-9.e+]+)", line)
                if m:
                    if m.group(1) == "TCC_HIT_sum": hit += float(m.group(2))
                    else: miss += float(m.group(2))
            if hit + miss > 0:
                row["l2_hit_rate_gemm_launches"] = round(hit / (hit + miss), 4)
    rows[name] = row
json.dump(dict(how="tools/energy_table.sh: one bench.py run per variant; ms_per_step from the timed region, socket power / sclk from rocm-smi over 2.5 s of the same "
                   "steps behind it (roofline.hwmon); joules_per_step = W x ms; L2 hit rate of the gemm_mx launches from a separate rocprofv3 --pmc pass",
               variants=rows), open(os.path.join(out, "energy.json"), "w"), indent=1)
for k, v in rows.items():
    print("%-26s %9.0f frames/s  %7.3f ms/step  %6.0f W  %5.3f GHz  %6.3f J/step  L2 hit %s" % (k, v["frames_per_s"], v["ms_per_step"], v["socket_power_W"] or 0, v["sclk_GHz"] or 0,
                                                                                              v["joules_per_step"] or 0, v.get("l2_hit_rate_gemm_launches", "-")))
PY
