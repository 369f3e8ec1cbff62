import torch, sys, os, subprocess, json
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from bench import synth
    from asvd4llm_amd import ops
    mats, scs = [], []
    for b in range(3):
        W, scal = synth(4096, 4096, 233 + b)
        mats.append(W.cuda()); scs.append(ops.make_scale(scal.cuda(), alpha=0.5))
    U, S, V, infos = ops.svd_batched(mats, scs, max_sweeps=1, want_vectors=False)
    torch.save([s.cpu() for s in S], sys.argv[1])
else:
    for maxd in (1, 1, 2, 2):
        out = []
        for pipe in (1, 1):
            env = dict(os.environ, ASVD_PIPE=str(pipe), ASVD_DBG_MAXD=str(maxd), ASVD_SPARSE="0")
            subprocess.run([sys.executable, "tools/r2_dbg8.py", f"/tmp/s_{len(out)}.pt"], env=env, check=True, stderr=subprocess.DEVNULL)
            out.append(torch.load(f"/tmp/s_{len(out)}.pt"))
        print("maxd", maxd, "max |S_piped - S_plain| per problem:", [f"{(a - b).abs().max().item():.3e}" for a, b in zip(out[0], out[1])])
