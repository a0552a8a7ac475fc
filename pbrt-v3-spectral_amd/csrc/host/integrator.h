// integrator.h -- host-side mirror of the reference's integrator surface for
// `Integrator "path"`: class Integrator { virtual void Render(const Scene&) = 0; }
// (src/core/integrator.h:53-58), SamplerIntegrator (integrator.h:77-113) and
// CreatePathIntegrator (src/integrators/path.h:69-71). Render() does not loop over
// tiles on the CPU: it hands the flat scene to the HIP path through the C ABI
// (include/mi_pt.h) and writes the film the way SamplerIntegrator::Render ends
// with camera->film->WriteImage() (integrator.cpp:341). Li() per ray is not
// offered on the host: the device evaluates Li for every camera sample inside
// the wavefront pipeline.
#pragma once
#include <string>
#include "scene.h"

namespace mipt {

class Integrator {
  public:
    virtual ~Integrator() {}
    // Returns 0 or a negative mi_status (the reference returns void and reports
    // through Error(); the C ABI needs a code). Message in *err.
    virtual int Render(const HostScene &scene, std::string *err) = 0;
};

class PathIntegrator : public Integrator {
  public:
    PathIntegrator(int deviceOrdinal, const std::string &outfile) : device(deviceOrdinal), outfile(outfile) {}
    int Render(const HostScene &scene, std::string *err) override;
    mi_counters counters{};
    double seconds = 0;
  private:
    int device;
    std::string outfile;
};

PathIntegrator *CreatePathIntegrator(const HostScene &scene, int deviceOrdinal, const std::string &outfile);

}  // namespace mipt
