"""How does the hipGraph executor run two independent chains?  Chains of spin kernels (torch.cuda._sleep: one thread, fixed duration, no
resources -- two of them can always run side by side), NA on the capture stream and NB on a second stream:
  (1) one graph, one stream (A then B);  (2) one graph with B forked onto the second stream and joined (what a model-level
  side stream becomes under capture);  (3) two graphs, one per chain, launched on two streams;  (4) eager, two streams.
Prints wall time per round; ideal concurrency = max(A, B), none = A + B.
usage: graph_fork_probe.py [NA] [NB] [spin-cycles]"""
import sys, time, torch
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 50
CYC = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
r = torch.zeros(16, device="cuda")


def chain(n):
    for _ in range(n):
        torch.cuda._sleep(CYC)


def timed(fn, rounds=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds * 1e6


# calibrate one chain
with torch.cuda.stream(s1):
    tA = timed(lambda: chain(NA))
print(f"eager chain of {NA} spins: {tA:.0f} us")

g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    torch.cuda.synchronize()
    with torch.cuda.graph(g1, stream=s1):
        r.add_(1); chain(NA); chain(NB); r.add_(1)
print(f"(1) one graph, one stream: {timed(g1.replay):.0f} us")

g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    torch.cuda.synchronize()
    with torch.cuda.graph(g2, stream=s1):
        r.add_(1)
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            chain(NB)
        chain(NA)
        s1.wait_stream(s2)
        r.add_(1)
print(f"(2) one graph, B forked to a second stream: {timed(g2.replay):.0f} us")

gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    torch.cuda.synchronize()
    with torch.cuda.graph(gA, stream=s1):
        chain(NA)
with torch.cuda.stream(s2):
    torch.cuda.synchronize()
    with torch.cuda.graph(gB, stream=s2):
        chain(NB)


def two_graphs():
    with torch.cuda.stream(s1):
        gA.replay()
    with torch.cuda.stream(s2):
        gB.replay()


print(f"(3) two graphs on two streams: {timed(two_graphs):.0f} us")


def eager2():
    with torch.cuda.stream(s1):
        chain(NA)
    with torch.cuda.stream(s2):
        chain(NB)


print(f"(4) eager, two streams: {timed(eager2):.0f} us")
