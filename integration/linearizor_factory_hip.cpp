// integration/linearizor_factory_hip.cpp - the one `case` a maintainer adds to the reference's factory
// (src/rootba/solver/linearizor.cpp:48-70), shown WITHOUT editing the reference tree:
//
//     case SolverOptions::SolverType::SQUARE_ROOT_HIP:                       // + the enum entry in solver_options.hpp
//       return std::make_unique<LinearizorHIP<Scalar>>(bal_problem, options, summary);
//
// oracle/build_ref.sh links with -Wl,--wrap on Linearizor<float|double>::create, so every call of the factory -
// in particular the one in the reference's LM loop, bal_bundle_adjustment.cpp:282 - lands in the functions below;
// they return the HIP linearizor when the square-root solver is requested and ROOTBA_LINEARIZOR=hip is set in the
// environment (the stand-in for the new enum entry), and otherwise call the reference's own factory.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "rootba/solver/linearizor.hpp"
#include "rootba/solver/linearizor_hip.hpp"

namespace {
template <class Scalar>
bool want_hip(const rootba::SolverOptions& options) {
  const char* e = std::getenv("ROOTBA_LINEARIZOR");
  return e && std::strcmp(e, "hip") == 0 && options.solver_type == rootba::SolverOptions::SolverType::SQUARE_ROOT;
}
}  // namespace

extern "C" {
// mangled names of rootba::Linearizor<Scalar>::create(BalProblem<Scalar>&, SolverOptions const&, SolverSummary*)
std::unique_ptr<rootba::Linearizor<double>> __real__ZN6rootba10LinearizorIdE6createERNS_10BalProblemIdEERKNS_13SolverOptionsEPNS_13SolverSummaryE(
    rootba::BalProblem<double>&, const rootba::SolverOptions&, rootba::SolverSummary*);
std::unique_ptr<rootba::Linearizor<float>> __real__ZN6rootba10LinearizorIfE6createERNS_10BalProblemIfEERKNS_13SolverOptionsEPNS_13SolverSummaryE(
    rootba::BalProblem<float>&, const rootba::SolverOptions&, rootba::SolverSummary*);

std::unique_ptr<rootba::Linearizor<double>> __wrap__ZN6rootba10LinearizorIdE6createERNS_10BalProblemIdEERKNS_13SolverOptionsEPNS_13SolverSummaryE(
    rootba::BalProblem<double>& bal_problem, const rootba::SolverOptions& options, rootba::SolverSummary* summary) {
  if (want_hip<double>(options)) return std::make_unique<rootba::LinearizorHIP<double>>(bal_problem, options, summary);
  return __real__ZN6rootba10LinearizorIdE6createERNS_10BalProblemIdEERKNS_13SolverOptionsEPNS_13SolverSummaryE(bal_problem, options, summary);
}
std::unique_ptr<rootba::Linearizor<float>> __wrap__ZN6rootba10LinearizorIfE6createERNS_10BalProblemIfEERKNS_13SolverOptionsEPNS_13SolverSummaryE(
    rootba::BalProblem<float>& bal_problem, const rootba::SolverOptions& options, rootba::SolverSummary* summary) {
  if (want_hip<float>(options)) return std::make_unique<rootba::LinearizorHIP<float>>(bal_problem, options, summary);
  return __real__ZN6rootba10LinearizorIfE6createERNS_10BalProblemIfEERKNS_13SolverOptionsEPNS_13SolverSummaryE(bal_problem, options, summary);
}
}
