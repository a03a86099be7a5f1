// One reference training iteration (train.py:159-190) as a native launch sequence:
// mask -> x[mask]=0 -> forward -> MSE on masked nodes -> backward -> Adam.  See gatres_train_step_t.
#include "gatres_common.h"
#include "gatres_layout.h"

extern "C" __attribute__((visibility("hidden"))) int gatres_adam_step_ex(float* params, const float* grads, float* exp_avg,
                                   float* exp_avg_sq, uint64_t* step_counter, int64_t count, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, const double* hp, float grad_scale,
                                   float* wt, int32_t num_blocks, int32_t nc, uint32_t* drop_count,
                                   const int32_t* mask_node_ptr, int32_t mask_graphs, double mask_rate, uint64_t mask_seed,
                                   uint8_t* mask_next, const uint64_t* mask_snap, void* stream);

// Parameter gradients, slab sum and (optionally) Adam of the fused path: the parameter-gradient launch followed by
// gatres_fused_finish.  (The one-launch form, gatres_fused_param_grads_finish, is slower on a single GPU -- 53 us against 38 +
// 9 inside the captured step, k_fused_host.hip -- and serves the data-parallel step's gradient buckets only.)
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_param_grads_ex(const gatres_model_t* m, const gatres_graph_t* g,
                                                                                 const float* saved, float* scratch,
                                                                                 const uint64_t* step_counter, void* stream);
extern "C" __attribute__((visibility("hidden"))) int gatres_fused_finish_ex(
    const gatres_model_t* m, const gatres_graph_t* g, float* scratch, float* grads, const float* loss_part, float* loss,
    int32_t do_adam, float* params, float* exp_avg, float* exp_avg_sq, uint64_t* step_counter, double lr, double beta1,
    double beta2, double eps, double weight_decay, const double* hparams, float grad_scale, const int32_t* mask_node_ptr,
    int32_t mask_graphs, double mask_rate, uint64_t mask_seed, uint8_t* mask_next, int32_t use_snap, void* stream);
static int fused_grads_and_update(const gatres_train_step_t* ts, const float* loss_part, bool adam, void* stream) {
  const bool want_next = adam && (ts->flags & GATRES_FLAG_MASK_NEXT) && ts->node_ptr && ts->num_graphs > 0 && ts->mask_next &&
                         ts->mask_next != ts->mask;
  // (the step count / fault snapshot: for this call's own update launch, or -- GATRES_FLAG_MASK_NEXT without the Adam phase:
  //  the data-parallel step -- for the sampling tail of the Adam phase that follows the all-reduce)
  const bool snap_later = !adam && (ts->flags & GATRES_FLAG_MASK_NEXT) && ts->step_counter;
  const int rc = gatres_fused_param_grads_ex(&ts->model, ts->graph, ts->saved, ts->scratch,
                                             (adam || snap_later) ? ts->step_counter : nullptr, stream);
  if (rc < 0 || rc > 1) return rc;
  // (rc == 1: the launch left the step count / fault word snapshot the sampling tail of the update launch reads; batches
  //  that leave CUs free form their parameter gradients on consumer workgroups instead -- no snapshot, no sampling ahead:
  //  the caller finds mask_next untouched... so it must not rely on it: GATResTrainer asks gatres_fused_finish_folds first)
  const bool mask_next = want_next && rc == 1;
  return gatres_fused_finish_ex(&ts->model, ts->graph, ts->scratch, ts->grads, loss_part, loss_part ? ts->loss : nullptr,
                                adam ? 1 : 0, ts->params, ts->exp_avg, ts->exp_avg_sq, ts->step_counter, ts->lr, ts->beta1,
                                ts->beta2, ts->eps, ts->weight_decay, ts->hparams, ts->grad_scale,
                                mask_next ? ts->node_ptr : nullptr, ts->num_graphs, ts->mask_rate, ts->seed,
                                mask_next ? ts->mask_next : nullptr, rc == 1 ? 1 : 0, stream);
}

extern "C" int gatres_train_step(const gatres_train_step_t* ts, void* stream) {
  if (!ts || !ts->graph || !ts->params || !ts->x || !ts->y || !ts->mask || !ts->out || !ts->g_out || !ts->loss ||
      !ts->saved || !ts->scratch)
    return GATRES_E_BADARG;
  const int N = ts->graph->num_nodes;
  int rc = 0;
  const bool fused = !(ts->flags & GATRES_FLAG_PER_OP) && gatres_fused_supported(&ts->model, ts->graph);
  // GATRES_FLAG_MASK_NEXT can only be honoured where the parameter gradients are a launch of their own (it leaves the step
  // count / fault snapshot the sampling tail reads).  Refused BEFORE anything is enqueued: a caller that asked for the next
  // mask and did not get it would train on a stale one without noticing.
  if (fused && (ts->flags & GATRES_FLAG_MASK_NEXT) && (ts->phases & GATRES_PHASE_ADAM) && (ts->phases & GATRES_PHASE_BACKWARD) &&
      ts->node_ptr && ts->num_graphs > 0 && ts->mask_next && ts->mask_next != ts->mask &&
      !gatres_fused_finish_folds(&ts->model, ts->graph))
    return GATRES_E_UNSUPPORTED;
  if ((ts->phases & GATRES_PHASE_MASK) && ts->node_ptr) {
    rc = gatres_mask_generate(ts->node_ptr, ts->num_graphs, ts->mask_rate, ts->seed, ts->step_counter, ts->mask,
                              stream);
    if (rc) return rc;
  }
  if (fused) {
    Layout L;
    if (!make_layout_g(&ts->model, ts->graph, &L)) return GATRES_E_UNSUPPORTED;
    float* loss_part = ts->scratch + L.sc_loss_part;
    const bool fwd = ts->phases & GATRES_PHASE_FORWARD, bwd = ts->phases & GATRES_PHASE_BACKWARD;
    const bool adam = ts->phases & GATRES_PHASE_ADAM;
    if ((bwd && !ts->grads) || (adam && (!ts->grads || !ts->exp_avg || !ts->exp_avg_sq || !ts->step_counter)))
      return GATRES_E_BADARG;
    // GATRES_FLAG_GRADS_DEFERRED: the backward phase stops after the chain (the kept g_h tables are in scratch); the caller
    // turns them into gradients itself, range by range (gatres_fused_param_grads_finish) -- the data-parallel step starts a
    // bucket's all-reduce between two such launches.
    const bool deferred = (ts->flags & GATRES_FLAG_GRADS_DEFERRED) != 0;
    if (ts->flags & GATRES_FLAG_GRADS_ONLY) {      // (reads the kept tables only: no transposed weights, nothing to prepare)
      if (!bwd || fwd || adam) return GATRES_E_BADARG;
      const bool top = ts->block_hi == ts->model.num_blocks;             // (the launch that also writes the loss)
      return gatres_fused_param_grads_finish(&ts->model, ts->graph, ts->saved, ts->scratch, ts->grads,
                                             top ? loss_part : nullptr, top ? ts->loss : nullptr, 0, nullptr, nullptr,
                                             nullptr, nullptr, 0., 0., 0., 0., 0., nullptr, 1.f, ts->block_lo, ts->block_hi,
                                             stream);
    }
    if (bwd && !(ts->flags & GATRES_FLAG_WT_VALID)) {      // (the fused Adam pass keeps scratch's W^T current)
      rc = gatres_fused_prepare_backward(&ts->model, ts->graph, ts->params, ts->scratch, stream);
      if (rc) return rc;
    }
    if (fwd && bwd) {
      rc = gatres_fused_run(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->y, ts->out, ts->g_out, loss_part,
                            nullptr, ts->saved, ts->scratch,
                            GATRES_PHASE_FORWARD | GATRES_PHASE_LOSS | GATRES_PHASE_BACKWARD, stream);
      if (rc || deferred) return rc;
      return fused_grads_and_update(ts, loss_part, adam, stream);
    }
    if (fwd) {
      rc = gatres_fused_run(&ts->model, ts->graph, ts->params, ts->x, ts->mask, nullptr, ts->out, nullptr, nullptr,
                            nullptr, ts->saved, ts->scratch, GATRES_PHASE_FORWARD, stream);
      if (rc) return rc;
      rc = gatres_masked_mse(ts->out, ts->y, ts->mask, ts->loss, ts->g_out, N, stream);
      if (rc) return rc;
    }
    if (bwd) {
      rc = gatres_fused_run(&ts->model, ts->graph, ts->params, ts->x, ts->mask, nullptr, nullptr, ts->g_out, nullptr,
                            nullptr, ts->saved, ts->scratch, GATRES_PHASE_BACKWARD, stream);
      if (rc || deferred) return rc;
      return fused_grads_and_update(ts, nullptr, adam, stream);
    }
    if (adam) {      // (the Adam-only phase of the data-parallel step: keeps scratch's transposed conv weights current too;
                     //  a step whose all-reduced gradient carries a fault mark is dropped and counted in status word 3;
                     //  GATRES_FLAG_MASK_NEXT: its extra workgroups sample the next step's mask, from the snapshot the
                     //  parameter-gradient launch of this step left in status words 8 .. 11)
      uint32_t* status = reinterpret_cast<uint32_t*>(ts->scratch + L.sc_flags + L.flag_words - 32);
      const bool next = (ts->flags & GATRES_FLAG_MASK_NEXT) && ts->node_ptr && ts->num_graphs > 0 && ts->mask_next &&
                        ts->mask_next != ts->mask;
      if (next && !gatres_fused_finish_folds(&ts->model, ts->graph)) return GATRES_E_UNSUPPORTED;      // (no snapshot was left)
      return gatres_adam_step_ex(ts->params, ts->grads, ts->exp_avg, ts->exp_avg_sq, ts->step_counter,
                                 gatres_param_count(ts->model.num_blocks, ts->model.nc), ts->lr, ts->beta1, ts->beta2,
                                 ts->eps, ts->weight_decay, ts->hparams, ts->grad_scale, ts->scratch + L.sc_wt,
                                 ts->model.num_blocks, ts->model.nc, status + 3, next ? ts->node_ptr : nullptr,
                                 ts->num_graphs, ts->mask_rate, ts->seed, next ? ts->mask_next : nullptr,
                                 reinterpret_cast<const uint64_t*>(status + 8), stream);
    }
    return 0;
  }
  if (ts->phases & GATRES_PHASE_FORWARD) {
    rc = gatres_model_forward_per_op(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->out, ts->saved, ts->scratch,
                              stream);
    if (rc) return rc;
    rc = gatres_masked_mse(ts->out, ts->y, ts->mask, ts->loss, ts->g_out, N, stream);
    if (rc) return rc;
  }
  if (ts->phases & GATRES_PHASE_BACKWARD) {
    if (!ts->grads) return GATRES_E_BADARG;
    rc = gatres_model_backward_per_op(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->g_out, ts->saved, ts->scratch,
                               ts->grads, nullptr, stream);
    if (rc) return rc;
  }
  if (ts->phases & GATRES_PHASE_ADAM) {
    if (!ts->grads || !ts->exp_avg || !ts->exp_avg_sq || !ts->step_counter) return GATRES_E_BADARG;
    rc = gatres_adam_step_ex(ts->params, ts->grads, ts->exp_avg, ts->exp_avg_sq, ts->step_counter,
                             gatres_param_count(ts->model.num_blocks, ts->model.nc), ts->lr, ts->beta1, ts->beta2, ts->eps,
                             ts->weight_decay, ts->hparams, ts->grad_scale, nullptr, 0, 0, nullptr, nullptr, 0, 0., 0,
                             nullptr, nullptr, stream);
    if (rc) return rc;
  }
  return 0;
}
