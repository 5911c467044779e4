"""Rewrite the measured rows of BASELINE.md / README.md from the bench JSON files under profiles/.

usage: python tools/update_results.py <headline bench json> [history note]
"""
import json
import re
import sys

ROOT = __file__.rsplit("/tools/", 1)[0]


def fmt(v):
    return f"{int(v):,}".replace(",", " ")


def main(headline, note=None):
    base = open(f"{ROOT}/BASELINE.md").read()
    readme = open(f"{ROOT}/README.md").read()
    labels = {"cnn10-fully": "10-node MNIST CNN fully connected", "femnist50-ring": "50-node FEMNIST CNN ring",
              "femnist1000-star": "1000-client FEMNIST CNN star (CFL)", "cnn100-encrypted-churn": None}
    for cfg, label in labels.items():
        d = json.load(open(f"{ROOT}/profiles/r1_bench_{cfg}.json"))
        r, v = d["rounds_per_sec"], d["value"]
        base = re.sub(r"(\| ours `bench.py --config " + re.escape(cfg) + r"` \| 1 \| )[\d.]+ \| [\d ]+ \|",
                      lambda m: f"{m.group(1)}{r:.2f} | {fmt(v)} |", base)
        if label:
            readme = re.sub(r"\| " + re.escape(label) + r" \| [\d.]+ \| [\d ]+ \|", f"| {label} | {r:.2f} | {fmt(v)} |", readme)
    d = json.load(open(headline))
    name = headline.rsplit("/", 1)[-1]
    base = re.sub(r"(hipGraph round\) \| 1 \| )[\d.]+ \| [\d ]+ \| `profiles/r1_v\d+_bench.json`",
                  lambda m: f"{m.group(1)}{d['rounds_per_sec']:.2f} | {fmt(d['value'])} | `profiles/{name}`", base)
    base = re.sub(r"\([\d.]+ x the eager loop\)", f"({d['vs_baseline']:.1f} x the eager loop)", base)
    if note:
        base = re.sub(r"(history: .*?)( \|\n)", lambda m: f"{m.group(1)}, {note}{m.group(2)}", base, count=1, flags=re.S)
    readme = re.sub(r"(\(headline\) \| )[\d.]+ \| [\d ]+ \|", lambda m: f"{m.group(1)}{d['rounds_per_sec']:.2f} | {fmt(d['value'])} |", readme)
    open(f"{ROOT}/BASELINE.md", "w").write(base)
    open(f"{ROOT}/README.md", "w").write(readme)


if __name__ == "__main__":
    main(*sys.argv[1:3])
