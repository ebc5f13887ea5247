import sys, json
sys.path.insert(0, "tools")
import batch_vs_sequential as bvs
r = bvs.kmeans_curves(2000, 10, minibatches=(2, 4, 8, 16, 32))
seq = r["sequential"]
for name in r:
    o = r[name]
    print("%-14s K %4d  tokens %6d  objective %12.3f  rel %+.4f   median sweep %.2f ms" % (name, o["components"][-1], o["n_tokens"][-1], o["sum_neg_len_sqrd_norm"][-1],
          (o["sum_neg_len_sqrd_norm"][-1] - seq["sum_neg_len_sqrd_norm"][-1]) / abs(seq["sum_neg_len_sqrd_norm"][-1]), 1e3 * sorted(o["sample_time"][1:])[len(o["sample_time"][1:]) // 2]))
json.dump(r, open("gpurun_out/r03m/minibatch_curves.json", "w"), indent=1)
