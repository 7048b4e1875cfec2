/*
 * rpt_frames.hpp — header-only C++ convenience over include/rpt.h for hosts that keep frames in flight.
 *
 * The reference finishes every frame before it starts the next (runKernel(), CLSetup.cpp:167-191).  One frame alone
 * leaves most of an MI355X idle (its critical path is the serial octree walk of its dearest pixel), so a host that
 * wants frame rate submits frame f into slot f mod N and presents the frame that slot finished N frames ago.  A slot
 * is an rpt_ctx of its own (stream, Object[], tile masks, framebuffer); all slots share ONE resident scene
 * (rpt_share_scene).  Nothing here is needed for the drop-in itself — it only spares the host the bookkeeping:
 *
 *     rpt::FrameRing ring(0, 3);                                   // device 0, three frames in flight
 *     ring.upload(desc); ring.set_params(wp, ambient, W, H, interval);
 *     for (;;) {                                                   // render()
 *         ... Lorentz update of cpu_objects ...
 *         void *finished = ring.acquire();                         // the frame submitted N calls ago, complete (or nullptr)
 *         if (finished) drawGL(finished);                          // consume it BEFORE its slot is given the next frame
 *         (void)ring.enqueue(&cpu_objects[0], (int)cpu_objects.size());   // returns at once; overwrites what acquire() returned
 *     }
 *     void *last = ring.drain();                                   // wait for everything; the newest frame
 *
 * acquire() and enqueue() are two calls on purpose: the framebuffer acquire() hands out belongs to the slot that the next
 * enqueue() renders into, so the host must be done with it (drawn, copied, or its own stream made to wait) before it
 * enqueues.  (An earlier revision returned the pointer FROM a call named submit(), i.e. after the overwriting launch was enqueued.)
 *
 * Every call returns/propagates the library's status through status(); a failed call leaves the ring usable.
 */
#ifndef RPT_FRAMES_HPP
#define RPT_FRAMES_HPP

#include <vector>

#include "rpt.h"

namespace rpt {

class FrameRing {
public:
    FrameRing(int device, int frames_in_flight) : status_(RPT_OK), next_(0), submitted_(0) {
        if (frames_in_flight < 1) frames_in_flight = 1;
        for (int k = 0; k < frames_in_flight && status_ == RPT_OK; k++) {
            rpt_ctx *c = nullptr;
            status_ = rpt_create(&c, device);
            if (status_ == RPT_OK) slots_.push_back(c);
        }
    }
    ~FrameRing() {
        for (rpt_ctx *c : slots_) rpt_destroy(c);
    }
    FrameRing(const FrameRing &) = delete;
    FrameRing &operator=(const FrameRing &) = delete;

    int status() const { return status_; }
    const char *last_error() const { return slots_.empty() ? "no context" : rpt_last_error(slots_[failed_slot_]); }
    int frames_in_flight() const { return (int)slots_.size(); }
    rpt_ctx *slot(int k) const { return slots_[(size_t)k]; }

    /* main.cpp:33-55: one upload, every other slot shares it */
    int upload(const rpt_scene_desc &desc) {
        if (slots_.empty()) return status_;
        if (!check(0, rpt_upload_scene(slots_[0], &desc))) return status_;
        for (size_t k = 1; k < slots_.size(); k++)
            if (!check(k, rpt_share_scene(slots_[k], slots_[0]))) return status_;
        return status_;
    }
    /* initCLKernel() / resize / interval toggle, for every slot; outputs are library-owned framebuffers */
    int set_params(const float white_point[3], float ambient, int width, int height, int interval) {
        for (size_t k = 0; k < slots_.size(); k++) {
            if (!check(k, rpt_set_params(slots_[k], white_point, ambient, width, height, interval))) return status_;
            if (!check(k, rpt_set_output(slots_[k], nullptr))) return status_;
        }
        return status_;
    }
    /* Wait for the frame the NEXT enqueue() will overwrite — the oldest one in flight — and return its framebuffer
     * (device pointer, 16 B/pixel), complete; nullptr while the ring is still filling or after an error.  The pointer
     * is valid until the next enqueue(). */
    void *acquire() {
        if (slots_.empty() || submitted_ < slots_.size()) return nullptr;
        if (!check(next_, rpt_sync(slots_[next_]))) return nullptr;
        return rpt_output_ptr(slots_[next_]);
    }
    /* Render.cpp:202-205 without the finish: refresh Object[] and enqueue the frame in the next slot; returns the
     * STATUS (0 = RPT_OK).  The slot's previous frame (what acquire() returned) is overwritten.
     * (Until round 2 a method called submit() returned the finished frame's pointer, or nullptr: `if (ring.submit(..)) present();`
     * written against that header would compile against a status-returning submit() and mean the opposite.  So the name is new,
     * and the old one is deleted: such a call site fails to compile instead of changing its meaning.) */
    [[nodiscard]] int enqueue(const void *objects, int count) {
        if (slots_.empty()) return status_;
        const size_t k = next_;
        if (!check(k, rpt_set_objects(slots_[k], objects, count))) return status_;
        if (!check(k, rpt_render_async(slots_[k]))) return status_;
        next_ = (next_ + 1) % slots_.size();
        submitted_++;
        return status_;
    }
    void *submit(const void *objects, int count) = delete;      /* see enqueue() */
    /* Wait for every frame in flight; returns the framebuffer of the newest one (nullptr if none was submitted). */
    void *drain() {
        for (size_t k = 0; k < slots_.size(); k++)
            if (!check(k, rpt_sync(slots_[k]))) return nullptr;
        if (!submitted_) return nullptr;
        return rpt_output_ptr(slots_[(next_ + slots_.size() - 1) % slots_.size()]);
    }
    /* the context that rendered the newest frame (for rpt_read_framebuffer after drain()) */
    rpt_ctx *newest() const { return slots_.empty() ? nullptr : slots_[(next_ + slots_.size() - 1) % slots_.size()]; }

private:
    bool check(size_t k, int rc) {
        status_ = rc;
        if (rc != RPT_OK) failed_slot_ = k;
        return rc == RPT_OK;
    }
    std::vector<rpt_ctx *> slots_;
    int status_;
    size_t next_, submitted_, failed_slot_ = 0;
};

}  // namespace rpt

#endif /* RPT_FRAMES_HPP */
