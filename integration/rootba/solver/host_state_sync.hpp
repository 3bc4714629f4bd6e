// integration/rootba/solver/host_state_sync.hpp - part of the reference-side binding (linearizor_hip.hpp).
#pragma once

namespace rootba {

// Host <-> device state protocol of the binding (see LinearizorHIP below): code that reads or writes BalProblem's
// LANDMARKS while a LinearizorHIP is alive (the reference's own LM loop does neither) brackets the access with these.
struct HostStateSync {
  virtual ~HostStateSync() = default;
  virtual void sync_host() = 0;           // bring BalProblem up to date with the device (before reading it)
  virtual void host_state_changed() = 0;  // BalProblem was modified by the caller: it is the source of truth again
};

}  // namespace rootba
