// The fly camera with the reference's class surface ("OpenGL Raytracer/camera.h":4-44).
// Window-system input (cursor, keys, clock) comes in through a plain struct instead of GLFW
// calls, so the same class drives an interactive front end, a fixed benchmark pose or a test.
#pragma once
#include "vecmath.h"

// What Camera::Update reads from GLFW in the reference (camera.cpp:28-48, 68-82).
struct CameraInput {
    double time = 0.0;                 // glfwGetTime()
    double cursorX = 0.0, cursorY = 0.0;  // glfwGetCursorPos()
    int windowWidth = 0, windowHeight = 0;  // glfwGetWindowSize()
    bool forward = false, backward = false, strafeRight = false, strafeLeft = false;  // W/Up, S/Down, D/Right, A/Left
};

class Camera {
    rth::mat4 viewMatrix;
    rth::mat4 projectionMatrix;
    rth::vec3 cameraPosition;
    rth::vec3 direction;
    rth::vec3 right;
    rth::vec3 up;
    float horizontalAngle;
    float verticalAngle;
    float initialFoV;
    float speed;
    float mouseSpeed;
    float scrollWheelY;
    bool firstTime;
    double lastTime;

public:
    bool cameraMouseControlEnabled;
    float aspectRatio;

    Camera();  // camera.cpp:6-18

    // camera.cpp:25-98.  With mouse control enabled the caller is expected to re-centre the
    // cursor afterwards, as the reference does with glfwSetCursorPos.
    void Update(const CameraInput &input);
    // Update with no input: a still camera at time 0 (what a benchmark run sees, main.cpp:422).
    void Update() { Update(CameraInput()); }

    rth::mat4 getViewMatrix() const { return viewMatrix; }
    rth::mat4 getProjectionMatrix() const { return projectionMatrix; }

    rth::vec3 getPosition() const { return cameraPosition; }
    rth::vec3 getDirection() const { return direction; }
    rth::vec3 getRight() const { return right; }
    rth::vec3 getUp() const { return up; }
    float getVerticalAngle() const { return verticalAngle; }
    float getHorizontalAngle() const { return horizontalAngle; }

    void setPosition(float x, float y, float z) { cameraPosition = rth::vec3(x, y, z); }
    void setPosition(rth::vec3 pos) { cameraPosition = pos; }
    void setVerticalAngle(float angle) { verticalAngle = angle; }
    void setHorizontalAngle(float angle) { horizontalAngle = angle; }
};

namespace rth {
// direction / right / up from the two angles, camera.cpp:50-65 (3.14f, not pi).
void camera_basis(float horizontalAngle, float verticalAngle, vec3 &direction, vec3 &right, vec3 &up);
}
