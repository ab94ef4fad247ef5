import sys, os, numpy as np, multiprocessing as mp
sys.path.insert(0, os.getcwd())
def owner(q_out, q_in):
    import torch, srcnn_cpp_amd as S
    ctx = S.Context(0)
    p = ctx.dev_alloc(1 << 20)
    data = (np.arange(1 << 20) % 251).astype(np.uint8)
    ctx.dev_upload(p, data)
    q_out.put(ctx.ipc_export(p))
    q_in.get()
    ctx.dev_free(p); ctx.close()
def reader(q_in, q_out):
    import torch, srcnn_cpp_amd as S
    ctx = S.Context(0)
    h = q_in.get()
    p = ctx.ipc_open(h)
    got = np.empty(1 << 20, np.uint8)
    ctx.dev_download(got, p)
    ok = bool((got == (np.arange(1 << 20) % 251).astype(np.uint8)).all())
    ctx.ipc_close(p)
    print("ipc read ok:", ok, flush=True)
    q_out.put(ok); ctx.close()
if __name__ == "__main__":
    mp.set_start_method("spawn")
    a, b = mp.Queue(), mp.Queue()
    o = mp.Process(target=owner, args=(a, b)); o.start()
    h = a.get()
    c, d = mp.Queue(), mp.Queue()
    r = mp.Process(target=reader, args=(c, d)); r.start()
    c.put(h)
    ok = d.get(timeout=120)
    b.put(1); o.join(); r.join()
    print("RESULT", ok, o.exitcode, r.exitcode)
