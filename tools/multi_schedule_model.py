"""Cost-model view of the round-6 launch schedule (no GPU needed): per-layer GEMM time of the routed decoder at the benchmark shape
under (a) one launch per GEMM, each paying ceil(tiles / 256) rounds of its own tile time, and (b) multi-problem launches with a
dynamic longest-first tile list (greedy list scheduling on 256 workgroups).  Tile time = 1.48 us per K tile + 5.3 us, launch
8.6 us (the fit of gemm_bf16.hip: cost256).      python tools/multi_schedule_model.py"""
import heapq

CU = 256
tile_us = lambda K: 1.48 * K / 64 + 5.3
tiles = lambda M, N: -(-M // 256) * -(-N // 256)


def single(M, N, K, groups=1):
    return 8.6 + -(-tiles(M, N) * groups // CU) * tile_us(K)


def multi(probs):
    """probs: (M, N, K, count) -> makespan of the longest-first dynamic list on CU workgroups"""
    items = []
    for M, N, K, cnt in probs:
        items += [tile_us(K)] * (tiles(M, N) * cnt)
    items.sort(reverse=True)
    heap = [0.0] * CU
    for d in items:
        heapq.heapreplace(heap, heap[0] + d)
    return 8.6 + max(heap), sum(items) / CU + 8.6


NL, NV, H, I, R, RG, KW = 11760, 4624, 4096, 11008, 1024, 2752, 4672
fwd = [("a  vision qkv A", [(NV, 3 * R + 64, H, 1)], None),
       ("F2 text qkv + 3 vision B", [(NL, 3 * H + 64, H, 1), (NV, H, R, 3)], [[0], [1]]),
       ("c  vision o A", [(NV, R, H, 1)], None),
       ("F4 text o + vision o B", [(NL, H, H, 1), (NV, H, R, 1)], [[0], [1]]),
       ("e  vision gate|up A", [(NV, 2 * RG, H, 1)], None),
       ("F6 text gate|up + 2 vision B", [(NL, 2 * I, H, 1), (NV, I, RG, 2)], [[0], [1]]),
       ("g  vision down A", [(NV, R, I, 1)], None),
       ("F8 text down + vision down B", [(NL, H, I, 1), (NV, H, R, 1)], [[0], [1]])]
bwd = [("dtd", [(NV, R, H, 1)], None),
       ("B1 dact_l + dact_v + dW down_B + dW down_A", [(NL, I, H, 1), (NV, I, R, 1), (H, R, KW, 1), (R, I, KW, 1)], [[0], [1], [2], [3]]),
       ("B2 dh2_l + 2 dtg + 2 dW gate/up_B", [(NL, H, 2 * I, 1), (NV, RG, I, 2), (I, RG, KW, 2)], [[0], [1], [2]]),
       ("B3 dh2_v + dW agu", [(NV, H, 2 * RG, 1), (2 * RG, H, KW, 1)], [[0], [1]]),
       ("dto", [(NV, R, H, 1)], None),
       ("B4 do_l + do_v + dW o_B + dW o_A", [(NL, H, H, 1), (NV, H, R, 1), (H, R, KW, 1), (R, H, KW, 1)], [[0], [1], [2], [3]]),
       ("B5 dh_l + 3 dt + 3 dW qkv_B", [(NL, H, 3 * H + 64, 1), (NV, R, H, 3), (H, R, KW, 3)], [[0], [1], [2]]),
       ("B6 dh_v + dW aqkv", [(NV, H, 3 * R + 64, 1), (3 * R + 64, H, KW, 1)], [[0], [1]])]

tot_s = tot_m = tot_i = 0.0
for name, probs, launches in fwd + bwd:
    if launches is None:
        s = m = single(*probs[0][:3], probs[0][3])
        ideal = m
    else:
        s = sum(single(*probs[i[0]][:3], probs[i[0]][3]) for i in launches)
        m, ideal = multi(probs)
    tot_s += s; tot_m += m; tot_i += ideal
    print(f"{name:48s} one-per-GEMM {s:8.1f} us   multi {m:8.1f} us   (perfectly packed {ideal:8.1f})   {m - s:+7.1f}")
print(f"{'per layer':48s} one-per-GEMM {tot_s:8.1f} us   multi {tot_m:8.1f} us   (perfectly packed {tot_i:8.1f})   {tot_m - tot_s:+7.1f} us "
      f"= {(tot_m - tot_s) * 32 / 1e3:+.1f} ms per 32-layer step")

# decoder_engine.CHAIN: the first low-rank stage joins the launch of its second stage (its tiles lead the list; the wait is free
# in this model because consumers sit at the end of the list)
chain = [("a + F2", fwd[0][1] + fwd[1][1]), ("c + F4", fwd[2][1] + fwd[3][1]), ("e + F6", fwd[4][1] + fwd[5][1]), ("g + F8", fwd[6][1] + fwd[7][1]),
         ("dtd + B1", bwd[0][1] + bwd[1][1]), ("B2", bwd[2][1]), ("B3", bwd[3][1]), ("dto + B4", bwd[4][1] + bwd[5][1]), ("B5", bwd[6][1]), ("B6", bwd[7][1])]
tot_c = sum(multi(p)[0] for _, p in chain)
print(f"{'per layer, chained first stages':48s} multi+chain {tot_c:8.1f} us   {tot_c - tot_s:+7.1f} us vs one-per-GEMM = {(tot_c - tot_s) * 32 / 1e3:+.1f} ms per step "
      f"({(tot_c - tot_m) * 32 / 1e3:+.1f} ms vs multi)")
