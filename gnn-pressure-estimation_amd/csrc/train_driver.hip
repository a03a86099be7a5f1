// One reference training iteration (train.py:159-190) as a native launch sequence:
// mask -> x[mask]=0 -> forward -> MSE on masked nodes -> backward -> Adam.  See gatres_train_step_t.
#include "gatres_common.h"
#include "gatres_layout.h"

extern "C" __attribute__((visibility("hidden"))) int gatres_adam_step_wt(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   uint64_t* step_counter, int64_t count, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, float grad_scale, float* wt, int32_t num_blocks,
                                   int32_t nc, void* stream);

extern "C" int gatres_train_step(const gatres_train_step_t* ts, void* stream) {
  if (!ts || !ts->graph || !ts->params || !ts->x || !ts->y || !ts->mask || !ts->out || !ts->g_out || !ts->loss ||
      !ts->saved || !ts->scratch)
    return GATRES_E_BADARG;
  const int N = ts->graph->num_nodes;
  int rc = 0;
  const bool fused = !(ts->flags & GATRES_FLAG_PER_OP) && gatres_fused_supported(&ts->model, ts->graph);
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
    if (bwd && !(ts->flags & GATRES_FLAG_WT_VALID)) {      // (the fused Adam pass keeps scratch's W^T current)
      rc = gatres_fused_prepare_backward(&ts->model, ts->graph, ts->params, ts->scratch, stream);
      if (rc) return rc;
    }
    if (fwd && bwd) {
      rc = gatres_fused_run(&ts->model, ts->graph, ts->params, ts->x, ts->mask, ts->y, ts->out, ts->g_out, loss_part,
                            nullptr, ts->saved, ts->scratch,
                            GATRES_PHASE_FORWARD | GATRES_PHASE_LOSS | GATRES_PHASE_BACKWARD, stream);
      if (rc) return rc;
      rc = gatres_fused_param_grads(&ts->model, ts->graph, ts->saved, ts->scratch, stream);
      if (rc) return rc;
      return gatres_fused_finish(&ts->model, ts->graph, ts->scratch, ts->grads, loss_part, ts->loss, adam ? 1 : 0,
                                 ts->params, ts->exp_avg, ts->exp_avg_sq, ts->step_counter, ts->lr, ts->beta1,
                                 ts->beta2, ts->eps, ts->weight_decay, ts->grad_scale, stream);
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
      if (rc) return rc;
      rc = gatres_fused_param_grads(&ts->model, ts->graph, ts->saved, ts->scratch, stream);
      if (rc) return rc;
      return gatres_fused_finish(&ts->model, ts->graph, ts->scratch, ts->grads, nullptr, nullptr, adam ? 1 : 0,
                                 ts->params, ts->exp_avg, ts->exp_avg_sq, ts->step_counter, ts->lr, ts->beta1,
                                 ts->beta2, ts->eps, ts->weight_decay, ts->grad_scale, stream);
    }
    if (adam)        // (the Adam-only phase of the data-parallel step: keeps scratch's transposed conv weights current too)
      return gatres_adam_step_wt(ts->params, ts->grads, ts->exp_avg, ts->exp_avg_sq, ts->step_counter,
                                 gatres_param_count(ts->model.num_blocks, ts->model.nc), ts->lr, ts->beta1, ts->beta2,
                                 ts->eps, ts->weight_decay, ts->grad_scale, ts->scratch + L.sc_wt, ts->model.num_blocks,
                                 ts->model.nc, stream);
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
    rc = gatres_adam_step(ts->params, ts->grads, ts->exp_avg, ts->exp_avg_sq, ts->step_counter,
                          gatres_param_count(ts->model.num_blocks, ts->model.nc), ts->lr, ts->beta1, ts->beta2, ts->eps,
                          ts->weight_decay, ts->grad_scale, stream);
    if (rc) return rc;
  }
  return 0;
}
