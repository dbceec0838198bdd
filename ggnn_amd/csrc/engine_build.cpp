// Engine behind the C-ABI, part "build": build / refine schedule of the shards of one GPU (graph_construction.cu:128-147).
// The handle is declared in engine.hpp.
#include "engine.hpp"

// GraphConstructionImpl::build / refine, graph_construction.cu:128-147, for all shards of ctx
void ggnn_handle::build_device(DeviceCtx& ctx, float tau_build, uint32_t refinement_iterations,
                  ggnn_measure measure)
{
  const uint32_t N = cfg.N, K = cfg.KBuild, KF = cfg.KF;
  hipStream_t stream = ctx.stream;
  // scratch (GraphBuffer, graph_buffer.cu:38-81); not overlapped -- HBM is plentiful
  DeviceBuffer nn1_dist(static_cast<size_t>(N) * 4), graph_buffer(static_cast<size_t>(N) * K * 4),
      rng(static_cast<size_t>(N) * 4), sym_buffer(static_cast<size_t>(N) * KF * 4),
      sym_atomic(static_cast<size_t>(N) * 4), stats_scratch(3 * kStatsBlocks * 4);
  ctx.build_ms = 0.f;
  uint64_t rng_calls = static_cast<uint64_t>(ctx.first_shard) << 16;
  // diagnostic mode (collect_counters): per-point work counters and an own HIP-event time of
  // every merge / sym launch, summed into build_work (ggnn_last_build_work)
  DeviceBuffer work;
  std::vector<uint32_t> h_work;
  hipEvent_t wev_a = nullptr, wev_b = nullptr;
  if (collect_counters) {
    work.alloc(static_cast<size_t>(N) * 16);
    h_work.resize(static_cast<size_t>(N) * 4);
    GGNN_HIP_CHECK(hipEventCreate(&wev_a));
    GGNN_HIP_CHECK(hipEventCreate(&wev_b));
  }
  struct EventGuard {
    hipEvent_t &a, &b;
    ~EventGuard()
    {
      if (a)
        (void)hipEventDestroy(a);
      if (b)
        (void)hipEventDestroy(b);
    }
  } event_guard{wev_a, wev_b};
  auto account = [&](ggnn_kernel_work& kw, uint32_t points, float ms) {
    GGNN_HIP_CHECK(hipMemcpy(h_work.data(), work.p, static_cast<size_t>(points) * 16,
                             hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lock(build_work_mutex);
    kw.launches += 1;
    kw.points += points;
    kw.ms += ms;
    for (uint32_t i = 0; i < points; ++i) {
      kw.n_dist += h_work[4 * i];
      kw.float_rows += h_work[4 * i + 1];
      kw.code_rows += h_work[4 * i + 2];
      kw.pops += h_work[4 * i + 3];
    }
  };

  for (uint32_t si = 0; si < ctx.shards.size(); ++si) {
    if (ctx.swap)  // out-of-core shards: this shard's rows into its slot, the pool is built in place
      acquire_shard(ctx, si, stream, /*with_graph=*/false);
    // the pre-screen copy serves the merge kernel too (made outside the timed region: it
    // depends on the base only and is kept for the queries)
    const bool use_ps = ensure_prescreen(ctx, si, measure);
    Shard& sh = ctx.shards[si];
    const void* base = shard_base(ctx, si);
    EventTimer timer(stream, ctx.ev_a, ctx.ev_b);

    auto layer_graph = [&](uint32_t l) {
      return sh.graph + static_cast<size_t>(cfg.Ns_offsets[l]) * K;
    };
    auto layer_tr = [&](uint32_t l) -> int32_t* {
      return l ? sh.translation + cfg.STs_offsets[l] : nullptr;
    };
    auto do_merge = [&](uint32_t top, uint32_t btm) {
      if (top == btm) {
        TopLaunch t{base,        base_dtype,          pad_D,
                    measure,     K,                   layer_tr(btm),
                    cfg.Ns[btm], btm ? cfg.S : cfg.S0, btm ? 0u : cfg.S0_off,
                    btm,         layer_graph(btm),    nn1_dist.as<float>()};
        launch_top(t, stream);
      }
      else {
        MergeLaunch m{base,
                      base_dtype,
                      measure,
                      cfg,
                      sh.graph,
                      sh.translation,
                      sh.selection,
                      sh.nn1_stats,
                      tau_build,
                      top,
                      btm,
                      graph_buffer.as<int32_t>(),
                      nn1_dist.as<float>(),
                      nullptr};
        if (use_ps) {
          m.ps_codes = sh.ps_codes.as<uint8_t>();
          m.ps_params = sh.ps_params.as<float>();
          m.ps_Dc = prescreen_code_dim(pad_D);
        }
        if (collect_counters) {
          m.n_work = work.as<uint32_t>();
          EventTimer t(stream, wev_a, wev_b);
          launch_merge(m, stream);
          account(build_work.merge, cfg.Ns[btm], t.stop());
        }
        else
          launch_merge(m, stream);
        GGNN_HIP_CHECK(hipMemcpyAsync(layer_graph(btm), graph_buffer.p,
                                      static_cast<size_t>(cfg.Ns[btm]) * K * 4,
                                      hipMemcpyDeviceToDevice, stream));
      }
      if (!btm)
        launch_nn1_stats(nn1_dist.as<float>(), N, stats_scratch.as<float>(), sh.nn1_stats,
                         stream);
    };
    auto do_select = [&](uint32_t layer) {
      if (!hook_rng.empty()) {
        GGNN_REQUIRE(hook_rng.size() >= static_cast<size_t>(layer + 1) * N, GGNN_INVALID_ARGUMENT,
                     "build hooks: rng needs (layers - 1) * N_shard numbers");
        GGNN_HIP_CHECK(hipMemcpyAsync(rng.p, hook_rng.data() + static_cast<size_t>(layer) * N,
                                      static_cast<size_t>(cfg.Ns[layer]) * 4,
                                      hipMemcpyHostToDevice, stream));
      }
      else
        launch_uniform(rng.as<float>(), cfg.Ns[layer], 1234ull, rng_calls++, stream);
      launch_select(cfg, layer, nn1_dist.as<float>(), rng.as<float>(), sh.translation,
                    sh.selection, stream);
    };
    auto do_sym = [&](uint32_t layer) {
      GGNN_HIP_CHECK(hipMemsetAsync(sym_buffer.p, 0xff,
                                    static_cast<size_t>(cfg.Ns[layer]) * KF * 4, stream));
      GGNN_HIP_CHECK(
          hipMemsetAsync(sym_atomic.p, 0, static_cast<size_t>(cfg.Ns[layer]) * 4, stream));
      SymLaunch s{base,
                  base_dtype,
                  measure,
                  pad_D,
                  K,
                  layer_graph(layer),
                  layer_tr(layer),
                  cfg.Ns[layer],
                  sh.nn1_stats,
                  tau_build,
                  sym_buffer.as<int32_t>(),
                  sym_atomic.as<uint32_t>(),
                  0,
                  cfg.Ns[layer]};
      // The short sym searches only gain from the pre-screen on wide rows (measured, 1M points:
      // D = 960 cosine 805 -> 454 ms per build, D = 128 74.6 -> 78.2 ms): used from 1 KB rows on.
      // Hook SYM_PRESCREEN = 0 | 1 forces it off / on (tuning hook).
      const int64_t sym_ps_hook = hook(kHookSymPrescreen);
      const bool sym_ps = sym_ps_hook >= 0 ? sym_ps_hook == 1 : pad_D >= 256;
      if (use_ps && sym_ps) {
        s.ps_codes = sh.ps_codes.as<uint8_t>();
        s.ps_params = sh.ps_params.as<float>();
        s.ps_Dc = prescreen_code_dim(pad_D);
      }
      if (hook_serial_sym) {
        // the reference's sym races through atomics and cross-block reads of sym_buffer
        // (sym_query_layer.cu:102-104 vs :133-136); one point per launch in ascending order is
        // the one schedule that is comparable with a CPU restatement
        for (uint32_t n = 0; n < cfg.Ns[layer]; ++n) {
          s.first_n = n;
          s.count = 1;
          launch_sym(s, stream);
        }
      }
      else if (collect_counters) {
        s.n_work = work.as<uint32_t>();
        EventTimer t(stream, wev_a, wev_b);
        launch_sym(s, stream);
        account(build_work.sym, cfg.Ns[layer], t.stop());
      }
      else
        launch_sym(s, stream);
      launch_sym_buffer_merge(K, cfg.Ns[layer], sym_buffer.as<int32_t>(),
                              sym_atomic.as<uint32_t>(), layer_graph(layer), stream);
    };

    // no selection/translation on layer 0; start from a defined state
    GGNN_HIP_CHECK(hipMemsetAsync(sh.translation, 0xff,
                                  2 * static_cast<size_t>(cfg.ST_all) * 4, stream));

    for (uint32_t top = 0; top < kLayers; ++top) {
      for (uint32_t btm = top; btm != 0xffffffffu; --btm) {
        do_merge(top, btm);
        if (top < kLayers - 1 && top == btm)
          do_select(top);
        do_sym(btm);
      }
    }
    for (uint32_t r = 0; r < refinement_iterations; ++r) {
      for (uint32_t layer = kLayers - 2; layer != 0xffffffffu; --layer) {
        do_merge(kLayers - 1, layer);
        do_sym(layer);
      }
    }
    const float ms = timer.stop();
    ctx.build_ms += ms;
    if (ctx.swap) {
      retire_built_shard(ctx, si);
      shard_consumed(ctx, si, stream);
    }
    sh.ready = true;
    GGNN_LOG(0, "[GPU: %d] build(): part %u => %.3f s [%u points -> %.3f us/point]", ctx.device,
             sh.global_id, ms / 1000.f, N, ms * 1000.f / static_cast<float>(N));
  }
  GGNN_HIP_CHECK(hipStreamSynchronize(stream));
}

void ggnn_handle::build(uint32_t KBuild, float tau_build, uint32_t refinement_iterations,
           ggnn_measure measure)
{
  prepare(KBuild);
  build_work = ggnn_build_work{};
  try {
    for_each_device(
        [&](DeviceCtx& ctx) { build_device(ctx, tau_build, refinement_iterations, measure); });
  }
  catch (...) {
    rollback_graph();
    throw;
  }
  build_ms = 0.f;
  for (const DeviceCtx& ctx : devs)
    build_ms += ctx.build_ms;  // "Sum of shard build times", ggnn.cu:237
  release_caller_copy();
}
