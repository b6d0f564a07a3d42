#include "Camera.h"

#include <cmath>

namespace rth {

void camera_basis(float h, float v, vec3 &direction, vec3 &right, vec3 &up) {
    direction = vec3(std::cos(v) * std::sin(h), std::sin(v), std::cos(v) * std::cos(h));
    right = vec3(std::sin(h - 3.14f / 2.0f), 0.0f, std::cos(h - 3.14f / 2.0f));
    up = cross(right, direction);
}

}  // namespace rth

Camera::Camera() {
    cameraPosition = rth::vec3(0.0f, 0.0f, 0.0f);
    horizontalAngle = 3.14f;
    verticalAngle = 0.0f;
    initialFoV = 90.0f;
    speed = 3.0f;
    mouseSpeed = 0.005f;
    scrollWheelY = 0.0f;
    firstTime = true;
    lastTime = 0.0;
    cameraMouseControlEnabled = true;
    aspectRatio = 4.0f / 3.0f;
}

void Camera::Update(const CameraInput &in) {
    if (firstTime) {
        lastTime = in.time;
        firstTime = false;
    }
    const float deltaTime = static_cast<float>(in.time - lastTime);

    if (cameraMouseControlEnabled && in.windowWidth > 0 && in.windowHeight > 0) {
        horizontalAngle += mouseSpeed * static_cast<float>(in.windowWidth / 2 - in.cursorX);
        verticalAngle += mouseSpeed * static_cast<float>(in.windowHeight / 2 - in.cursorY);
    }
    rth::camera_basis(horizontalAngle, verticalAngle, direction, right, up);

    if (in.forward) cameraPosition += direction * deltaTime * speed;
    if (in.backward) cameraPosition -= direction * deltaTime * speed;
    if (in.strafeRight) cameraPosition += right * deltaTime * speed;
    if (in.strafeLeft) cameraPosition -= right * deltaTime * speed;

    const float FoV = initialFoV - 1.0f * scrollWheelY;
    projectionMatrix = rth::perspective(FoV, aspectRatio, 0.1f, 150.0f);
    viewMatrix = rth::look_at(cameraPosition, cameraPosition + direction, up);
    lastTime = in.time;
}
