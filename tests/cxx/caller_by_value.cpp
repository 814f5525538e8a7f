// A caller of the reference interface compiled against include/ ALONE: holds VolumeRenderCL by
// value like the reference's widget (/root/reference/src/qt/volumerenderwidget.h:225) and walks
// the reference's call order (volumerenderwidget.cpp:242 initialize, :808 loadVolumeData,
// :942 setTransferFunction, :647 updateOutputImg, :1098 updateView, :478 runRaycastNoGL).
// tests/test_boundary_compile.py compiles (and links) it; it is never run without a GPU.
#include <volumerendercl.h>

#include <cstdio>

struct Widget {
    VolumeRenderCL _volumerender;        // by value
    std::vector<float> _outputData;
};

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    Widget w;
    try {
        w._volumerender.initialize(false, false);
        DatRawReader::Properties props;
        props.dat_file_name = argv[1];
        const size_t timesteps = w._volumerender.loadVolumeData(props);
        std::vector<unsigned char> tff(1024 * 4, 0);
        for (size_t i = 0; i < 1024; ++i) tff[4 * i + 3] = (unsigned char)(i / 4);
        w._volumerender.setTransferFunction(tff);
        std::vector<unsigned int> prefix(1024);
        unsigned int acc = 0;
        for (size_t i = 0; i < 1024; ++i) prefix[i] = (acc += tff[4 * i + 3]);
        w._volumerender.setTffPrefixSum(prefix);
        w._volumerender.updateOutputImg(64, 48, 0);
        const std::array<float, 16> view = {{2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 2, 0, 0, 0, 1}};
        w._volumerender.updateView(view);
        w._volumerender.updateSamplingRate(1.5);
        w._volumerender.runRaycastNoGL(64, 48, w._outputData);
        const std::array<unsigned int, 4> res = w._volumerender.getResolution();
        std::printf("%zu timesteps, %ux%ux%u, %zu floats, %.6f s\n", timesteps, res[0], res[1], res[2],
                    w._outputData.size(), w._volumerender.getLastExecTime());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return w._outputData.size() == 64u * 48u * 4u ? 0 : 3;
}
