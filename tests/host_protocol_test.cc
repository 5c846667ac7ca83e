// Exercises rasr_amd/host/*.hh against librasr_amd.so on a GPU (driven by tests/test_host_gpu.py).
// Prints one line per check; the Python side compares the scores with the oracle.
#define AMXHOST_REQUIRE_THROWS
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../rasr_amd/host/BatchFeatureScorer.hh"
#include "../rasr_amd/host/MfccNode.hh"

using namespace AmxHost;

static int fails = 0;
#define CHECK(c)                                             \
    do {                                                     \
        if (!(c)) {                                          \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                         \
        }                                                    \
    } while (0)

int main(int argc, char** argv) {
    amx_ctx* ctx = nullptr;
    if (amx_init(0, &ctx) != AMX_OK) {
        printf("FAIL init: %s\n", amx_last_error());
        return 2;
    }
    // ---- a small GMM, 3 mixtures x 2 densities, d = 4
    const int             dim = 4, M = 3;
    std::vector<uint32_t> off = {0, 2, 4, 6}, didx = {0, 1, 2, 3, 4, 5}, dmean = {0, 1, 2, 3, 4, 5}, dcov(6, 0);
    std::vector<double>   lw(6, std::log(0.5));
    std::vector<float>    means(6 * dim), var(dim, 1.5f);
    std::mt19937          rng(5);
    std::normal_distribution<float> nd;
    for (auto& v : means)
        v = nd(rng);
    amx_gmm_model model = {dim, M, 6, 6, 1, off.data(), didx.data(), lw.data(), dmean.data(), dcov.data(), means.data(), var.data(), 1.0, 1.0};

    const unsigned     B = 4;
    BatchFeatureScorer fs(std::unique_ptr<BatchBackend>(new GmmBackend(ctx, model)), B);
    CHECK(fs.isBuffered() && fs.bufferSize() == B && fs.bufferEmpty() && !fs.bufferFilled());
    CHECK(fs.nMixtures() == 3 && fs.dimension() == 4);

    // reference scores for T frames, computed in one plain batch call
    const int          T = 11;
    std::vector<float> feats(T * dim), want(T * M);
    for (auto& v : feats)
        v = nd(rng);
    {
        GmmBackend direct(ctx, model);
        CHECK(direct.score(feats.data(), T, want.data()) == AMX_OK);
    }
    // the caller protocol of Speech/Recognizer.cc: every frame must come back once, in order
    std::vector<Scorer> fed;
    for (int t = 0; t < T; ++t) {
        FeatureVector f(feats.begin() + t * dim, feats.begin() + (t + 1) * dim);
        if (fs.isBuffered() && !fs.bufferFilled())
            fs.addFeature(f);
        else
            fed.push_back(fs.getScorer(f));
        // the decoder consumes the scorer right away
        if (!fed.empty()) {
            int k = (int)fed.size() - 1;
            for (int e = 0; e < M; ++e)
                CHECK(fed[k]->score(e) == want[k * M + e]);
        }
    }
    while (!fs.bufferEmpty()) {
        fed.push_back(fs.flush());
        int k = (int)fed.size() - 1;
        for (int e = 0; e < M; ++e)
            CHECK(fed[k]->score(e) == want[k * M + e]);
    }
    CHECK((int)fed.size() == T);
    CHECK(fed[0]->nEmissions() == 3);
    // resident score block: every frame's row crossed PCIe exactly once (T rows of M scores), nothing else
    CHECK(fs.bytesToHost() == (size_t)T * M * sizeof(float));
    {
        // lazy rows and the gather extension: after a refill only the rows / pairs that are asked for come back
        BatchFeatureScorer lazy(std::unique_ptr<BatchBackend>(new GmmBackend(ctx, model)), B);
        Scorer first;
        for (int t = 0; t < (int)B; ++t) {
            FeatureVector f(feats.begin() + t * dim, feats.begin() + (t + 1) * dim);
            if (!lazy.bufferFilled())
                lazy.addFeature(f);
            else
                first = lazy.getScorer(f);
        }
        CHECK(first && lazy.bytesToHost() == 0);                   // nothing scored, nothing copied yet
        const EmissionIndex want_e[2] = {2, 0};
        Score               got[2];
        first->scores(want_e, 2, got);                             // frame 0: two scores through the device gather
        CHECK(got[0] == want[0 * M + 2] && got[1] == want[0 * M + 0]);
        CHECK(lazy.bytesToHost() == 2 * sizeof(float));
        Scorer second = lazy.flush();                              // frame 1
        CHECK(second->score(1) == want[1 * M + 1]);                 // first score() of a frame fetches its row
        CHECK(lazy.bytesToHost() == 2 * sizeof(float) + M * sizeof(float));
        CHECK(second->score(2) == want[1 * M + 2]);                 // ... and the next one is a host read
        CHECK(lazy.bytesToHost() == 2 * sizeof(float) + M * sizeof(float));
        second->scores(want_e, 2, got);                            // a fetched row also serves the list call
        CHECK(got[0] == want[1 * M + 2] && lazy.bytesToHost() == 2 * sizeof(float) + M * sizeof(float));
    }
    // contract violations
    bool threw = false;
    try {
        fs.flush();
    } catch (const std::logic_error&) {
        threw = true;
    }
    CHECK(threw);  // require(!bufferEmpty())
    threw = false;
    try {
        fs.getScorer(FeatureVector(dim, 0.f));
    } catch (const std::logic_error&) {
        threw = true;
    }
    CHECK(threw);  // require(bufferFilled())
    fs.reset();
    CHECK(fs.bufferEmpty());
    for (unsigned i = 0; i + 1 < B; ++i)
        fs.addFeature(FeatureVector(dim, 0.f));
    CHECK(fs.bufferFilled());
    threw = false;
    try {
        fs.addFeature(FeatureVector(dim, 0.f));
    } catch (const std::logic_error&) {
        threw = true;
    }
    CHECK(threw);  // require(!bufferFilled())

    // ---- NN backend in the bench's default arithmetic (AMX_PREC_F16MX): a buffer fill whose features leave the f16 range must FAIL
    // ITSELF (scoreResident waits for the pass, amx_ffnn_wait_dev) -- the decoder never sees a row of it -- not the fill after it
    {
        const int    nin = 8, nout = 5, Tn = 6;
        std::vector<float> W((size_t)nout * nin), bias(nout, 0.1f);
        for (size_t i = 0; i < W.size(); ++i)
            W[i] = 0.01f * (float)((int)(i % 7) - 3);
        const float* Wp[1]   = {W.data()};
        const float* bp[1]   = {bias.data()};
        const int    ind[1]  = {nin}, outd[1] = {nout}, act[1] = {AMX_ACT_NONE};
        amx_ffnn_model nm;
        memset(&nm, 0, sizeof nm);
        nm.n_layers = 1; nm.in_dim = ind; nm.out_dim = outd; nm.W = Wp; nm.bias = bp; nm.activation = act; nm.precision = AMX_PREC_F16MX;
        FfnnBackend nb(ctx, nm);
        std::vector<float> x((size_t)Tn * nin, 0.5f), row(nout);
        CHECK(nb.scoreResident(x.data(), Tn) == AMX_OK);
        CHECK(nb.fetchRow(2, row.data()) == AMX_OK && std::isfinite(row[0]));
        x[3 * nin + 1] = 1.0e6f;
        CHECK(nb.scoreResident(x.data(), Tn) == AMX_ERR_STATE && strstr(amx_last_error(), "f16 range"));   // THIS fill fails
        unsigned r0 = 0, e0 = 0;
        float    one = 0.f;
        CHECK(nb.fetchPairs(1, &r0, &e0, &one) != AMX_OK);                                                  // ... and exposes no row
    }

    // ---- MFCC node: parameters, attributes, packets, timestamps
    MfccNode node(ctx);
    CHECK(node.setParameter("nr-outputs", "12") && node.setParameter("filter-width", "268.258") && !node.setParameter("bogus", "1"));
    std::map<std::string, std::string> attr;
    attr["sample-rate"] = "16000";
    CHECK(node.configure(attr));
    CHECK(node.outputAttributes().at("sample-rate") == "1" && node.outputAttributes().at("frame-shift") == "0.01" &&
          node.outputAttributes().at("datatype") == "vector-f32");
    std::vector<float> pcm(16000);
    for (size_t i = 0; i < pcm.size(); ++i)
        pcm[i] = 8000.f * std::sin(0.05f * i) + 100.f * nd(rng);
    node.putSamples(pcm.data(), 4096, 2.5);
    node.putSamples(pcm.data() + 4096, pcm.size() - 4096, 2.5 + 4096 / 16000.0);
    CHECK(node.eos());
    FeaturePacket p;
    int           n = 0;
    double        lastEnd = 0;
    FILE*         dump = argc > 1 ? fopen(argv[1], "wb") : nullptr;
    while (node.getFeature(p)) {
        CHECK(p.data.size() == 12);
        if (n == 0)
            CHECK(p.startTime == 2.5 && std::fabs(p.endTime - 2.525) < 1e-12);
        if (dump)
            fwrite(p.data.data(), 4, 12, dump);
        lastEnd = p.endTime;
        ++n;
    }
    if (dump) {
        fwrite(pcm.data(), 4, pcm.size(), dump);
        fclose(dump);
    }
    CHECK(n == 99);                                   // ceil((16000-400)/160)+1
    CHECK(std::fabs(lastEnd - (2.5 + 1.0)) < 1e-9);   // last (short) frame ends with the audio
    // the same segment as a vector-s16 stream (node behind the audio reader, no converter node): integer-valued samples give
    // the same bits on both paths
    {
        std::vector<int16_t> pcm16(16000);
        std::vector<float>   pcmf(16000);
        for (size_t i = 0; i < pcm16.size(); ++i) {
            pcm16[i] = (int16_t)std::lrint(pcm[i]);
            pcmf[i]  = (float)pcm16[i];
        }
        std::vector<std::vector<float>> a, b;
        node.putSamples(pcmf.data(), pcmf.size(), 0.0);
        CHECK(node.eos());
        while (node.getFeature(p))
            a.push_back(p.data);
        node.putSamples(pcm16.data(), 1000, 0.0);
        node.putSamples(pcm16.data() + 1000, pcm16.size() - 1000, 1000 / 16000.0);
        CHECK(node.eos());
        while (node.getFeature(p))
            b.push_back(p.data);
        CHECK(a.size() == 99 && b.size() == 99);
        bool same = a.size() == b.size();
        for (size_t i = 0; same && i < a.size(); ++i)
            same = memcmp(a[i].data(), b[i].data(), 12 * sizeof(float)) == 0;
        CHECK(same);
    }
    attr["sample-rate"] = "0";
    CHECK(!node.configure(attr) && strstr(amx_last_error(), "not positive"));

    amx_destroy(ctx);
    printf(fails ? "FAILED %d\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
